"""CPU emulator of the C ABI in include/segnb_hip.h.  TEST INFRASTRUCTURE ONLY.

Each method restates, with torch-CPU ops on raw host memory, what the same-named entry point of
libsegnb_hip.so is specified to compute.  Uses:
  * tests/test_plan_cpu.py injects it (segnb._native.set_backend_for_testing) to run the product's HOST
    logic -- tap tables, channel maps, concat wiring, backward routing, flat params -- on CPU against
    the oracle (no GPU needed);
  * tests/test_hip_ops.py (-m gpu) compares every HIP kernel with the matching method on random inputs.
The product never imports this file and has no CPU path.
"""
import ctypes
import math
import os

import numpy as np
import torch

F32, BF16 = 0, 1
REPL = 16      # SEGNB_STAT_REPLICAS
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _tdt(code):
    return torch.bfloat16 if code == BF16 else torch.float32


def _mem(ptr, numel, dtype):
    """torch view of `numel` elements of raw host memory at address `ptr`."""
    if isinstance(ptr, ctypes.c_void_p):
        ptr = ptr.value
    nbytes = numel * torch.empty((), dtype=dtype).element_size()
    buf = (ctypes.c_char * nbytes).from_address(int(ptr))
    return torch.frombuffer(buf, dtype=dtype)


def _nhwc(ptr, N, H, W, C, ld, dtype):
    """[N,H,W,C] strided view (pixel stride ld) of raw memory."""
    npix = N * H * W
    flat = _mem(ptr, (npix - 1) * ld + C, dtype)
    return torch.as_strided(flat, (N, H, W, C), (H * W * ld, W * ld, ld, 1))


def _geom(g):
    if hasattr(g, 'contents'):
        g = g.contents
    return g


def _ints(p, n):
    if p is None:
        return None
    if isinstance(p, (list, tuple)):
        return list(p)
    return [int(p[i]) for i in range(n)]


def _gather(X, g, t):
    """[N,QH,QW,Ci] fp32: input pixels (qh*in_step+dh[t], qw*in_step+dw[t]), zero outside."""
    N, Hi, Wi, Ci = X.shape
    hi = torch.arange(g.QH) * g.in_step + g.dh[t]
    wi = torch.arange(g.QW) * g.in_step + g.dw[t]
    mh = (hi >= 0) & (hi < Hi)
    mw = (wi >= 0) & (wi < Wi)
    sub = X[:, hi.clamp(0, Hi - 1)][:, :, wi.clamp(0, Wi - 1)].float()
    mask = (mh[:, None] & mw[None, :]).to(sub.dtype)
    return sub * mask[None, :, :, None]


class AbiEmulator(object):
    # ------------------------------------------------------------------------------------------ conv
    def segnb_conv_fprop(self, g, dtype, in_p, wp, bias, bias_n, out_p, stats, stream):
        g = _geom(g)
        dt = _tdt(dtype)
        X = _nhwc(in_p, g.N, g.Hi, g.Wi, g.Ci, g.ld_in, dt)
        O = _nhwc(out_p, g.N, g.Ho, g.Wo, g.Co, g.ld_out, dt)
        Wm = _mem(wp, g.Co * g.ntaps * g.Ci, dt).view(g.Co, g.ntaps, g.Ci).float()
        acc = torch.zeros(g.N, g.QH, g.QW, g.Co)
        for t in range(g.ntaps):
            acc += _gather(X, g, t) @ Wm[:, t, :].t()
        if bias is not None and bias_n > 0:
            b = torch.zeros(g.Co)
            b[:bias_n] = _mem(bias, bias_n, torch.float32)
            acc += b
        stored = acc.to(dt)
        oh = torch.arange(g.QH) * g.out_step + g.oh0
        ow = torch.arange(g.QW) * g.out_step + g.ow0
        O[:, oh[:, None], ow[None, :], :] = stored
        if stats is not None:
            S = _mem(stats, REPL * 2 * g.Co, torch.float64).view(REPL, 2, g.Co)[0]
            v = stored.double().reshape(-1, g.Co)
            S[0] += v.sum(0)
            S[1] += (v * v).sum(0)
        return 0

    # ---- uint8 HWC input (NormalizeImage, lib/augmentations.py:452-460, on the way in)
    @staticmethod
    def _norm_u8(img, N, H, W, C, scale, mean, std):
        u = _mem(img, N * H * W * C, torch.uint8).view(N, H, W, C).float()
        m = torch.tensor([float(mean[i]) for i in range(C)])
        inv = 1.0 / torch.tensor([float(std[i]) for i in range(C)], dtype=torch.float32)
        return (u * torch.tensor(scale, dtype=torch.float32) - m) * inv

    def segnb_pack_input_u8(self, img, N, H, W, C, scale, mean, std, out, dtype, Cp, ld, stream):
        O = _nhwc(out, N, H, W, Cp, ld, _tdt(dtype))
        O.zero_()
        O[..., :C] = self._norm_u8(img, N, H, W, C, scale, mean, std).to(_tdt(dtype))
        return 0

    def segnb_conv_fprop_u8_ok(self, g, dtype):
        g = _geom(g)
        return int(dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.Ci == 8 and g.Co <= 32
                   and g.Wo >= 12 and g.Hi == g.Ho and g.Wi == g.Wo and g.QH == g.Ho and g.QW == g.Wo)

    def segnb_conv_fprop_u8(self, g, img, C, scale, mean, std, wp, bias, bias_n, out_p, stats, x_packed, ld_packed, stream):
        gg = _geom(g)
        x = torch.zeros(gg.N, gg.Hi, gg.Wi, 8, dtype=torch.bfloat16)
        x[..., :C] = self._norm_u8(img, gg.N, gg.Hi, gg.Wi, C, scale, mean, std).to(torch.bfloat16)
        if x_packed is not None:
            _nhwc(x_packed, gg.N, gg.Hi, gg.Wi, 8, ld_packed, torch.bfloat16).copy_(x)
        keep = x.contiguous()
        g2 = type(gg)()
        ctypes.memmove(ctypes.addressof(g2), ctypes.addressof(gg), ctypes.sizeof(gg))
        g2.ld_in = 8
        return self.segnb_conv_fprop(g2, BF16, keep.data_ptr(), wp, bias, bias_n, out_p, stats, stream)

    # ---- convolution with the affine + activation epilogue (segnb_conv_fprop_act)
    def segnb_conv_fprop_act(self, g, dtype, in_p, wp, bias, bias_n, out_p, ep, stream):
        gg, e = _geom(g), _geom(ep)
        dt = _tdt(dtype)
        X = _nhwc(in_p, gg.N, gg.Hi, gg.Wi, gg.Ci, gg.ld_in, dt)
        O = _nhwc(out_p, gg.N, gg.Ho, gg.Wo, gg.Co, gg.ld_out, dt)
        Wm = _mem(wp, gg.Co * gg.ntaps * gg.Ci, dt).view(gg.Co, gg.ntaps, gg.Ci).float()
        acc = torch.zeros(gg.N, gg.QH, gg.QW, gg.Co)
        for t in range(gg.ntaps):
            acc += _gather(X, gg, t) @ Wm[:, t, :].t()
        b = torch.zeros(gg.Co)
        if bias is not None and bias_n > 0:
            b[:bias_n] = _mem(bias, bias_n, torch.float32)
        if e.coef:
            co = _mem(e.coef, 4 * gg.Co, torch.float32).view(4, gg.Co)
            v = acc * co[0] + ((b - co[2]) * co[0] + co[1])
        else:
            v = acc + b
        neg = 0.0 if e.act == 1 else (float(e.slope) if e.act == 2 else 1.0)
        v = torch.where(v < 0, v * neg, v) + 0.0
        oh = torch.arange(gg.QH) * gg.out_step + gg.oh0
        ow = torch.arange(gg.QW) * gg.out_step + gg.ow0
        O[:, oh[:, None], ow[None, :], :] = v.to(dt)
        return 0

    # ---- data gradient + the BatchNorm-backward reduction of its output's producer (segnb_conv_fprop_bnreduce) = the two
    # separate entry points, composed
    # ---- virtual concat: cat([Upsample x2(u), skip]) materialised here, then the plain entry point
    def segnb_conv_upcat_ok(self, g, dtype, Cu):
        g = _geom(g)
        return int(g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.Hi % 2 == 0 and g.Wi % 2 == 0 and 0 < Cu < g.Ci)

    def _upcat(self, g, dtype, in_p, src):
        gg, sc = _geom(g), _geom(src)
        dt = _tdt(dtype)
        Cs = gg.Ci - sc.Cu
        U = _nhwc(sc.u, gg.N, gg.Hi // 2, gg.Wi // 2, sc.Cu, sc.ld_u, dt)
        S = _nhwc(in_p, gg.N, gg.Hi, gg.Wi, Cs, gg.ld_in, dt)
        cat = torch.cat([U.repeat_interleave(2, 1).repeat_interleave(2, 2), S], 3).contiguous()
        g2 = type(gg)()
        ctypes.memmove(ctypes.addressof(g2), ctypes.addressof(gg), ctypes.sizeof(gg))
        g2.ld_in = gg.Ci
        return g2, cat

    def segnb_conv_fprop_upcat(self, g, dtype, in_p, src, wp, bias, bias_n, out_p, stats, stream):
        g2, cat = self._upcat(g, dtype, in_p, src)
        return self.segnb_conv_fprop(g2, dtype, cat.data_ptr(), wp, bias, bias_n, out_p, stats, stream)

    def segnb_conv_wgrad_upcat(self, g, dtype, in_p, src, dout_p, dwp, nslab, stream):
        g2, cat = self._upcat(g, dtype, in_p, src)
        return self.segnb_conv_wgrad(g2, dtype, cat.data_ptr(), dout_p, dwp, nslab, stream)

    # ---- data gradient with the Upsample(x2) backward in its store pass: the plain entry point into a scratch copy, then
    # the first Cu channels leave as 2 x 2 sums of the STORED (rounded) values, the others as they are
    def segnb_conv_fprop_upsum_ok(self, g, dtype, Cu):
        g = _geom(g)
        return int(dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.QH == g.Ho and g.QW == g.Wo and
                   g.Ho % 2 == 0 and g.Wo % 2 == 0 and g.Wo >= 12 and g.Ci % 32 == 0 and g.Ci <= 96 and g.Co <= 96 and
                   g.Co % 8 == 0 and Cu % 8 == 0 and 0 < Cu < g.Co)

    def segnb_conv_fprop_upsum(self, g, dtype, in_p, wp, out_p, dst, stream):
        gg, d = _geom(g), _geom(dst)
        dt = _tdt(dtype)
        full = torch.zeros(gg.N, gg.Ho, gg.Wo, gg.Co, dtype=dt)
        g2 = type(gg)()
        ctypes.memmove(ctypes.addressof(g2), ctypes.addressof(gg), ctypes.sizeof(gg))
        g2.ld_out = gg.Co
        self.segnb_conv_fprop(g2, dtype, in_p, wp, None, 0, full.data_ptr(), None, stream)
        O = _nhwc(out_p, gg.N, gg.Ho, gg.Wo, gg.Co, gg.ld_out, dt)
        O[..., d.Cu:] = full[..., d.Cu:]
        DU = _nhwc(d.u, gg.N, gg.Ho // 2, gg.Wo // 2, d.Cu, d.ld_u, dt)
        f = full[..., :d.Cu].float()
        DU.copy_((((f[:, 0::2, 0::2] + f[:, 0::2, 1::2]) + f[:, 1::2, 0::2]) + f[:, 1::2, 1::2]).to(dt))
        return 0

    def segnb_upconv_fprop_acc_ok(self, N, H, W, Ci, Co, ld_out, dtype):
        return 1                                            # (every shape: the segmented forward plan is exercised)

    def segnb_upconv_fprop_acc(self, dtype, N, H, W, Ci, ld_in, in_p, wp, Co, CoW, out_p, ld_out, stats, stream):
        """out[n, 2Y+py, 2X+px] += sum_taps u[n, Y+dh, X+dw] . W[phase][:, tap, :], tap lists of convt_fwd(4, 2, 1)"""
        from segnb import convplan as cpl
        dt = _tdt(dtype)
        U = _nhwc(in_p, N, H, W, Ci, ld_in, dt).float()
        O = _nhwc(out_p, N, 2 * H, 2 * W, Co, ld_out, dt)
        Wm = _mem(wp, 4 * CoW * 4 * Ci, dt).view(4, CoW, 4, Ci).float()
        _, launches, full = cpl.convt_fwd(H, W, 4, 4, 2, 1)
        assert full and len(launches) == 4
        Up = torch.nn.functional.pad(U, (0, 0, 1, 1, 1, 1))
        res = O.float().clone()
        for ph, l in enumerate(launches):
            acc = torch.zeros(N, H, W, Co)
            for t, (dh, dw, _, _) in enumerate(l.taps):
                acc += Up[:, 1 + dh:1 + dh + H, 1 + dw:1 + dw + W, :] @ Wm[ph, :Co, t, :].t()
            res[:, l.oh0::2, l.ow0::2, :] += acc
        stored = res.to(dt)
        O.copy_(stored)
        if stats is not None:
            S = _mem(stats, REPL * 2 * Co, torch.float64).view(REPL, 2, Co)[0]
            v = stored.double().reshape(-1, Co)
            S[0] += v.sum(0)
            S[1] += (v * v).sum(0)
        return 0

    def segnb_upconv_fprop_ok(self, N, H, W, Ci, Co, ld_out, dtype):
        return int(dtype == BF16 and Ci % 64 == 0 and Ci >= 128 and W >= 12)

    def segnb_upconv_fprop_act(self, dtype, N, H, W, Ci, ld_in, in_p, wp, Co, CoW, bias, bias_n, out_p, ld_out, ep, stream):
        e = _geom(ep)
        assert not e.coef
        return self.segnb_upconv_fprop(dtype, N, H, W, Ci, ld_in, in_p, wp, Co, CoW, bias, bias_n, out_p, ld_out, None, stream,
                                       _act=(e.act, e.slope))

    def segnb_upconv_fprop(self, dtype, N, H, W, Ci, ld_in, in_p, wp, Co, CoW, bias, bias_n, out_p, ld_out, stats, stream, _act=None):
        """out = bias + the four phase sums (ConvTranspose2d(4, 2, 1) forward): zero, then the accumulating form, bias folded
        into the first rounding"""
        dt = _tdt(dtype)
        O = _nhwc(out_p, N, 2 * H, 2 * W, Co, ld_out, dt)
        from segnb import convplan as cpl
        U = _nhwc(in_p, N, H, W, Ci, ld_in, dt).float()
        Wm = _mem(wp, 4 * CoW * 4 * Ci, dt).view(4, CoW, 4, Ci).float()
        _, launches, full = cpl.convt_fwd(H, W, 4, 4, 2, 1)
        assert full and len(launches) == 4
        Up = torch.nn.functional.pad(U, (0, 0, 1, 1, 1, 1))
        b = torch.zeros(Co)
        if bias is not None and bias_n > 0:
            b[:bias_n] = _mem(bias, bias_n, torch.float32)
        res = torch.zeros(N, 2 * H, 2 * W, Co)
        for ph, l in enumerate(launches):
            acc = torch.zeros(N, H, W, Co)
            for t, (dh, dw, _, _) in enumerate(l.taps):
                acc += Up[:, 1 + dh:1 + dh + H, 1 + dw:1 + dw + W, :] @ Wm[ph, :Co, t, :].t()
            res[:, l.oh0::2, l.ow0::2, :] = acc + b
        if _act is not None:
            if _act[0] == ACT_RELU:
                res = torch.relu(res)
            elif _act[0] == ACT_LEAKY:
                res = torch.nn.functional.leaky_relu(res, _act[1])
        stored = res.to(dt)
        O.copy_(stored)
        if stats is not None:
            S = _mem(stats, REPL * 2 * Co, torch.float64).view(REPL, 2, Co)[0]
            v = stored.double().reshape(-1, Co)
            S[0] += v.sum(0)
            S[1] += (v * v).sum(0)
        return 0

    def segnb_conv_fprop_upd_ok(self, g, dtype):
        g = _geom(g)
        return int(g.ntaps == 16 and g.in_step == 2)       # (the emulator serves every geometry: the segmented plan is exercised)

    def segnb_conv_fprop_bnreduce_ok(self, g, dtype):
        g = _geom(g)
        if not (dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.QH == g.Ho and g.QW == g.Wo
                and g.Co % 8 == 0):
            return 0
        if g.Ci <= 24 and g.Co >= 32:          # a dense layer's data gradient (growth 16 -> the prefix): the general kernel's store pass
            return 1
        if g.Wo < 12:
            return 0
        return int(g.Ci % 32 == 0 and g.Ci <= 96 and g.Co <= 64 and not (g.Co > 32 and g.Ci > 32))

    def segnb_conv_fprop_bnapply_ok(self, g, dtype):
        g = _geom(g)
        return int(dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.QH == g.Ho and g.QW == g.Wo
                   and g.Co % 8 == 0 and g.Ci <= 24 and g.Co > 32)

    def _dgrad_tmp(self, g, dtype, in_p, wp):
        """the (rounded) data gradient of the two-launch forms, in a dense temporary; -> (tensor kept alive, pointer, stride)"""
        gg = _geom(g)
        tmp = torch.zeros(gg.N * gg.Ho * gg.Wo * gg.Co, dtype=_tdt(dtype))
        g2 = self._regeom(gg, ld_out=gg.Co)
        rc = self.segnb_conv_fprop(g2, dtype, in_p, wp, None, 0, tmp.data_ptr(), None, 0)
        assert rc == 0
        return tmp, gg

    def segnb_conv_fprop_bnsums(self, g, dtype, in_p, wp, ep, stream):
        tmp, gg = self._dgrad_tmp(g, dtype, in_p, wp)
        e = _geom(ep)
        return self.segnb_bn_act_bwd_reduce(dtype, e.y, e.ld_y, gg.N, gg.Ho, gg.Wo, gg.Co, e.coef, e.act, e.slope, None,
                                            tmp.data_ptr(), gg.Co, None, 0, None, 0, None, 0, e.sums, None, 0, stream)

    def segnb_conv_fprop_bnapply(self, g, dtype, in_p, wp, ep, stream):
        tmp, gg = self._dgrad_tmp(g, dtype, in_p, wp)
        e = _geom(ep)
        fn = self.segnb_bn_bwd_apply_fused_direct_acc if e.accumulate else self.segnb_bn_bwd_apply_fused_direct
        return fn(dtype, e.y, e.ld_y, gg.N, gg.Ho, gg.Wo, e.C, gg.Co, e.coef, e.sums, e.gamma, e.bcoef, e.dgamma, e.dbeta, 1, None,
                  e.act, e.slope, tmp.data_ptr(), gg.Co, e.dx, e.ld_dx, stream)

    def segnb_conv_fprop_drop_ok(self, g, dtype):
        g = _geom(g)
        dh, dw = [g.dh[t] for t in range(g.ntaps)], [g.dw[t] for t in range(g.ntaps)]
        s1 = (g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.oh0 == 0 and g.ow0 == 0 and g.QH == g.Ho and g.QW == g.Wo
              and g.Wo > 8 and max(dh) - min(dh) == 2 and max(dw) - min(dw) == 2)
        M, Ktot = g.N * g.QH * g.QW, g.ntaps * g.Ci
        general_blocks = -(-M // 128) * -(-g.Co // (32 if g.Co <= 32 else 64))
        deepk = g.Ci % 16 == 0 and Ktot >= 2048 and general_blocks * 2 <= 256
        return int(dtype == BF16 and g.Co <= 32 and g.Co % 8 == 0 and g.Ci > 96 and g.Ci % 8 == 0 and (s1 or deepk))

    def segnb_conv_fprop_drop(self, g, dtype, in_p, wp, bias, bias_n, out_p, dropmul, ld_drop, stats, stats_ld, stream):
        """segnb_conv_fprop, the Dropout2d multipliers on the stored (rounded) result, the statistics of what that leaves"""
        gg = _geom(g)
        rc = self.segnb_conv_fprop(g, dtype, in_p, wp, bias, bias_n, out_p, None, stream)
        if rc:
            return rc
        dt = _tdt(dtype)
        O = _nhwc(out_p, gg.N, gg.Ho, gg.Wo, gg.Co, gg.ld_out, dt)
        m = _mem(dropmul, gg.N * ld_drop, torch.float32).view(gg.N, ld_drop)[:, :gg.Co]
        O.copy_((O.float() * m[:, None, None, :]).to(dt))
        if stats:
            return self.segnb_bn_stats_ld(dtype, out_p, gg.ld_out, gg.N, gg.Ho, gg.Wo, gg.Co, stats, stats_ld, stream)
        return 0

    def segnb_conv_fprop_actmask_ok(self, g, dtype):
        g = _geom(g)
        if not (dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.QH == g.Ho and g.QW == g.Wo
                and g.oh0 == 0 and g.ow0 == 0 and g.Co % 8 == 0):
            return 0
        if g.Ci == 32 and g.Co <= 32 and g.Wo >= 32:          # conv_roll_kernel
            return 1
        # the MASK instantiation of conv_fprop_ws_kernel: 64-channel input chunks, > 32 output channels, a row-major 3 x 3 window
        dh, dw = [g.dh[t] for t in range(9)], [g.dw[t] for t in range(9)]
        window = (all(dh[t] == dh[3 * (t // 3)] and dw[t] == dw[t % 3] for t in range(9))
                  and sorted(set(d - min(dh) for d in dh)) == [0, 1, 2] and sorted(set(d - min(dw) for d in dw)) == [0, 1, 2])
        return int(g.Ci % 64 == 0 and g.Co > 32 and g.Wo > 8 and window)

    def segnb_conv_fprop_bnreduce(self, g, dtype, in_p, wp, out_p, ep, stream):
        gg, e = _geom(g), _geom(ep)
        rc = self.segnb_conv_fprop(g, dtype, in_p, wp, None, 0, out_p, None, stream)
        if rc:
            return rc
        if not e.coef:          # activation mask of a layer without BatchNorm: out becomes dz
            return self.segnb_bn_act_bwd_reduce(dtype, e.y, e.ld_y, gg.N, gg.Ho, gg.Wo, gg.Co, None, e.act, e.slope, None,
                                                out_p, gg.ld_out, None, 0, None, 0, out_p, gg.ld_out, e.sums, None, 0, stream)
        return self.segnb_bn_act_bwd_reduce(dtype, e.y, e.ld_y, gg.N, gg.Ho, gg.Wo, gg.Co, e.coef, e.act, e.slope, None,
                                            out_p, gg.ld_out, None, 0, None, 0, None, 0, e.sums, None, 0, stream)

    # ---- operands recomputed on load (segnb_operand_tf): materialise the operand with the two-launch form, then the plain op
    def segnb_conv_fprop_tf_ok(self, g, dtype, kind):
        g = _geom(g)
        return int(dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.QH == g.Ho and g.QW == g.Wo
                   and g.Hi == g.Ho and g.Wi == g.Wo and g.Ci == 32 and g.Co <= 32 and kind in (1, 2))

    def segnb_conv_wgrad_tf_ok(self, g, dtype):
        return self.segnb_conv_fprop_tf_ok(g, dtype, 1)

    def _tf_operand(self, tf, src, ld_src, N, H, W, C, dtype):
        """-> dense [N,H,W,C] tensor of `dtype`: the operand a segnb_operand_tf describes"""
        tf = _geom(tf)
        assert tf.Cp == C, 'the emulator takes transforms over exactly the channels of the operand'
        tmp = torch.zeros(N * H * W * C, dtype=_tdt(dtype))
        if tf.kind == 1:
            self.segnb_bn_act_fwd(dtype, src, ld_src, N, H, W, C, tf.coef, tf.act, tf.slope, tf.drop, tmp.data_ptr(), C,
                                  None, 0, None, 0, None, 0, 0)
        else:
            assert tf.kind == 2 and not tf.drop
            self.segnb_bn_bwd_apply_direct(dtype, tf.y, tf.ld_y, N, H, W, C, tf.coef, tf.bcoef, tf.act, tf.slope, src, ld_src,
                                           tmp.data_ptr(), C, None, C, 0)
        return tmp

    @staticmethod
    def _regeom(g, **kw):
        g2 = type(g)()
        ctypes.memmove(ctypes.addressof(g2), ctypes.addressof(g), ctypes.sizeof(g))
        for k, v in kw.items():
            setattr(g2, k, v)
        return g2

    def segnb_conv_fprop_tf(self, g, dtype, in_p, tf, wp, bias, bias_n, out_p, stats, bn, stream):
        gg = _geom(g)
        tmp = self._tf_operand(tf, in_p, gg.ld_in, gg.N, gg.Hi, gg.Wi, gg.Ci, dtype)
        g2 = self._regeom(gg, ld_in=gg.Ci)
        rc = self.segnb_conv_fprop(g2, dtype, tmp.data_ptr(), wp, bias, bias_n, out_p, stats, stream)
        if rc or not bn:
            return rc
        e = _geom(bn)
        return self.segnb_bn_act_bwd_reduce(dtype, e.y, e.ld_y, gg.N, gg.Ho, gg.Wo, gg.Co, e.coef, e.act, e.slope, None,
                                            out_p, gg.ld_out, None, 0, None, 0, None, 0, e.sums, None, 0, stream)

    def segnb_conv_wgrad_tf(self, g, dtype, in_p, tf_in, dout_p, tf_dout, dwp, nslab, stream):
        gg = _geom(g)
        kw, keep = {}, []
        if tf_in:
            keep.append(self._tf_operand(tf_in, in_p, gg.ld_in, gg.N, gg.Hi, gg.Wi, gg.Ci, dtype))
            in_p, kw['ld_in'] = keep[-1].data_ptr(), gg.Ci
        if tf_dout:
            keep.append(self._tf_operand(tf_dout, dout_p, gg.ld_out, gg.N, gg.Ho, gg.Wo, gg.Co, dtype))
            dout_p, kw['ld_out'] = keep[-1].data_ptr(), gg.Co
        return self.segnb_conv_wgrad(self._regeom(gg, **kw), dtype, in_p, dout_p, dwp, nslab, stream)

    # slab count of the emulated device: stride-1 3x3 bf16 launches write EMU_SLABS partial slabs (the HIP library
    # derives its count from the CU count); everything else accumulates into one zeroed slab
    EMU_SLABS = 3

    def segnb_conv_wgrad_slabs(self, g, dtype):
        g = _geom(g)
        s1 = (g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.oh0 == 0 and g.ow0 == 0 and g.QH == g.Ho
              and g.QW == g.Wo and max(g.dh) - min(g.dh) == 2 and max(g.dw) - min(g.dw) == 2)
        return self.EMU_SLABS if (s1 and dtype == 1) else 1

    def segnb_conv_wgrad(self, g, dtype, in_p, dout_p, dwp, nslab, stream):
        if nslab != self.segnb_conv_wgrad_slabs(g, dtype):
            return 1
        g = _geom(g)
        dt = _tdt(dtype)
        X = _nhwc(in_p, g.N, g.Hi, g.Wi, g.Ci, g.ld_in, dt)
        D = _nhwc(dout_p, g.N, g.Ho, g.Wo, g.Co, g.ld_out, dt)
        oh = torch.arange(g.QH) * g.out_step + g.oh0
        ow = torch.arange(g.QW) * g.out_step + g.ow0
        d = D[:, oh[:, None], ow[None, :], :].float().reshape(-1, g.Co)
        G = _mem(dwp, nslab * g.Co * g.ntaps * g.Ci, torch.float32).view(nslab, g.Co, g.ntaps, g.Ci)
        if nslab > 1:
            G[0].zero_()                   # slabs are overwritten, the result is slab 0 ...
            G[1:].fill_(float('nan'))      # ... and the others are scratch: poison them
        for t in range(g.ntaps):
            G[0, :, t, :] += d.t() @ _gather(X, g, t).reshape(-1, g.Ci)
        tgt, self._wg_target = getattr(self, '_wg_target', None), None
        if tgt is not None:
            # segnb_wgrad_target: the result goes to the parameter's own gradient instead of staying in slab 0
            assert tgt['ntaps'] == g.ntaps
            co = torch.arange(tgt['Co'])[:, None, None]
            ci = torch.arange(tgt['Ci'])[None, None, :]
            kp = torch.tensor(tgt['kpos'], dtype=torch.long)[None, :, None]
            idx = (co * tgt['s_out'] + (tgt['ci_off'] + ci) * tgt['s_in'] + kp).reshape(-1)
            dst = _mem(tgt['gw'], int(idx.max()) + 1, torch.float32)
            src = G[0, :tgt['Co'], :, :tgt['Ci']].reshape(-1)
            if tgt['accumulate']:
                dst.index_add_(0, idx, src)
            else:
                dst[idx] = src
            G.fill_(float('nan')) if nslab > 1 else G.zero_()      # scratch afterwards (one slab: left zeroed for the atomics)
        return 0

    def segnb_wgrad_target_arm(self, t):
        t = _geom(t)
        self._wg_target = dict(gw=int(t.gw), s_out=int(t.s_out), s_in=int(t.s_in), ci_off=int(t.ci_off), Ci=int(t.Ci), Co=int(t.Co),
                               accumulate=int(t.accumulate), ntaps=int(t.ntaps), kpos=[int(t.kpos[i]) for i in range(int(t.ntaps))])
        return 0

    def segnb_conv_wgrad_bnapply_ok(self, g, dtype):
        """(the library's rule: by default only where the first layer's rolling kernel serves -- 8 padded input channels, rows of
        at least 32 pixels -- SEGNB_WGRAD_BNAPPLY=1 widens it to every thin stride-1 3x3 layer, =0 disables)"""
        g = _geom(g)
        e = os.environ.get('SEGNB_WGRAD_BNAPPLY')
        if e == '0' or not (dtype == BF16 and g.ntaps == 9 and g.in_step == 1 and g.out_step == 1 and g.Co <= 32 and g.Co % 8 == 0):
            return 0
        same = g.Hi == g.Ho and g.Wi == g.Wo and g.QH == g.Ho and g.QW == g.Wo and g.oh0 == 0 and g.ow0 == 0
        if g.Ci == 8 and g.Wo >= 32 and same and getattr(self, 'tuned', {}).get('wgrad_c8roll', 1):
            return 1
        return int(e is not None)

    def segnb_conv_wgrad_bnapply(self, g, dtype, in_p, gsrc, ld_g, y, ld_y, coef, bcoef, Cp, act, slope, dwp, nslab, stream):
        """segnb_bn_bwd_apply_direct into a temporary, then segnb_conv_wgrad on it"""
        gg = _geom(g)
        tmp = torch.zeros(gg.N * gg.Ho * gg.Wo * Cp, dtype=_tdt(dtype))
        self.segnb_bn_bwd_apply_direct(dtype, y, ld_y, gg.N, gg.Ho, gg.Wo, Cp, coef, bcoef, act, slope, gsrc, ld_g,
                                       tmp.data_ptr(), Cp, None, gg.Co, stream)
        g2 = type(gg)()
        ctypes.memmove(ctypes.addressof(g2), ctypes.addressof(gg), ctypes.sizeof(gg))
        g2.ld_out = Cp
        return self.segnb_conv_wgrad(g2, dtype, in_p, tmp.data_ptr(), dwp, nslab, stream)

    def _maps(self, Mp, Cp, ntaps, tap_off, mmap, cmap):
        mm = _mem(mmap, Mp, torch.int32).long()
        cm = _mem(cmap, Cp, torch.int32).long()
        to = torch.tensor(_ints(tap_off, ntaps), dtype=torch.long)
        return mm, cm, to

    def segnb_pack_weight(self, w, wp, dtype, Mp, Cp, ntaps, s_m, s_c, tap_off, mmap, cmap, stream):
        mm, cm, to = self._maps(Mp, Cp, ntaps, tap_off, mmap, cmap)
        idx = mm.clamp(min=0)[:, None, None] * s_m + to[None, :, None] + cm.clamp(min=0)[None, None, :] * s_c
        valid = (mm >= 0)[:, None, None] & (cm >= 0)[None, None, :]
        valid = valid.expand(Mp, ntaps, Cp)
        n = int(idx[valid].max()) + 1 if valid.any() else 1
        src = _mem(w, n, torch.float32)
        vals = torch.where(valid, src[idx.clamp(max=n - 1)], torch.zeros(()))
        _mem(wp, Mp * ntaps * Cp, _tdt(dtype)).view(Mp, ntaps, Cp).copy_(vals.to(_tdt(dtype)))
        return 0

    def segnb_unpack_wgrad(self, dwp, gw, Mp, Cp, ntaps, s_m, s_c, tap_off, mmap, cmap, accumulate, stream):
        mm, cm, to = self._maps(Mp, Cp, ntaps, tap_off, mmap, cmap)
        idx = mm.clamp(min=0)[:, None, None] * s_m + to[None, :, None] + cm.clamp(min=0)[None, None, :] * s_c
        valid = ((mm >= 0)[:, None, None] & (cm >= 0)[None, None, :]).expand(Mp, ntaps, Cp)
        src = _mem(dwp, Mp * ntaps * Cp, torch.float32).view(Mp, ntaps, Cp)
        n = int(idx[valid].max()) + 1
        dst = _mem(gw, n, torch.float32)
        if accumulate:
            dst.index_add_(0, idx[valid], src[valid])
        else:
            dst[idx[valid]] = src[valid]
        src.zero_()
        return 0

    # batched forms: decode the device job table (segnb.engine.PACK_JOB_DTYPE) and run the single-job methods
    JOB_BYTES = 336

    def segnb_pack_job_bytes(self):
        return self.JOB_BYTES

    def segnb_pack_job_blocks(self, Mp, Cp, ntaps, s_m, s_c):
        if min(s_m, s_c) > 9 or ntaps > 64:
            return -1
        if s_c < s_m:
            return (Cp + 255) // 256 * Mp
        return (Mp + 7) // 8 * ((Cp + 63) // 64)

    def _jobs(self, jobs, njobs):
        raw = bytes((ctypes.c_char * (njobs * self.JOB_BYTES)).from_address(int(jobs)))
        dt = np.dtype([('w', '<u8'), ('packed', '<u8'), ('mmap', '<u8'), ('cmap', '<u8'), ('s_m', '<i8'),
                       ('s_c', '<i8'), ('Mp', '<i4'), ('Cp', '<i4'), ('ntaps', '<i4'), ('dtype', '<i4'),
                       ('block_start', '<i4'), ('nslab', '<i4'), ('masked', '<i4'), ('pad_', '<i4'),
                       ('tap_off', '<i4', (64,))])
        return np.frombuffer(raw, dtype=dt)

    # masked jobs (PackJob.masked): tap_off[t] is a bit mask over the <= 9 kernel positions of the parameter --
    # pack: packed(m, t, c) = sum_{k in mask[t]} w(m, c, k), rounded once;  unpack: gw(m, c, k) += sum_{t: k in mask[t]} dwp(m, t, c)
    def _pack_masked(self, j):
        Mp, Cp, nt = int(j['Mp']), int(j['Cp']), int(j['ntaps'])
        dt = _tdt(int(j['dtype']))
        acc = torch.zeros(Mp, nt, Cp)
        for t in range(nt):
            for k in range(9):
                if int(j['tap_off'][t]) >> k & 1:
                    one = torch.zeros(Mp * Cp, dtype=torch.float32)
                    self.segnb_pack_weight(int(j['w']), one.data_ptr(), F32, Mp, Cp, 1, int(j['s_m']), int(j['s_c']), [k],
                                           int(j['mmap']), int(j['cmap']), 0)
                    acc[:, t, :] += one.view(Mp, Cp)
        _mem(int(j['packed']), Mp * nt * Cp, dt).view(Mp, nt, Cp).copy_(acc.to(dt))

    def _unpack_masked(self, j):
        Mp, Cp, nt = int(j['Mp']), int(j['Cp']), int(j['ntaps'])
        src = _mem(int(j['packed']), Mp * nt * Cp, torch.float32).view(Mp, nt, Cp)
        for k in range(9):
            part = torch.zeros(Mp, 1, Cp)
            hit = False
            for t in range(nt):
                if int(j['tap_off'][t]) >> k & 1:
                    part[:, 0, :] += src[:, t, :]
                    hit = True
            if hit:
                keep = part.clone()
                self.segnb_unpack_wgrad(keep.data_ptr(), int(j['w']), Mp, Cp, 1, int(j['s_m']), int(j['s_c']), [k],
                                        int(j['mmap']), int(j['cmap']), 1, 0)
        src.zero_()

    def segnb_pack_weight_multi(self, jobs, njobs, total_blocks, stream):
        for j in self._jobs(jobs, njobs):
            if int(j['masked']):
                self._pack_masked(j)
                continue
            self.segnb_pack_weight(int(j['w']), int(j['packed']), int(j['dtype']), int(j['Mp']), int(j['Cp']),
                                   int(j['ntaps']), int(j['s_m']), int(j['s_c']),
                                   [int(v) for v in j['tap_off'][:int(j['ntaps'])]], int(j['mmap']), int(j['cmap']), stream)
        return 0

    def segnb_unpack_wgrad_multi(self, jobs, njobs, total_blocks, stream):
        for j in self._jobs(jobs, njobs):
            ns = int(j['nslab'])
            if ns > 1:                      # partial slabs (job field nslab > 1): slab 0 += slabs 1.., in order
                n1 = int(j['Mp']) * int(j['ntaps']) * int(j['Cp'])
                G = _mem(int(j['packed']), ns * n1, torch.float32).view(ns, n1)
                for sl in range(1, ns):
                    G[0] += G[sl]
            if int(j['masked']):
                self._unpack_masked(j)
                continue
            self.segnb_unpack_wgrad(int(j['packed']), int(j['w']), int(j['Mp']), int(j['Cp']), int(j['ntaps']),
                                    int(j['s_m']), int(j['s_c']), [int(v) for v in j['tap_off'][:int(j['ntaps'])]],
                                    int(j['mmap']), int(j['cmap']), 1, stream)
        return 0

    # both matrices of a plain 3x3 convolution from one job record (segnb_pack_weight_pair_multi)
    PAIR_JOB = np.dtype([('w', '<u8'), ('pf', '<u8'), ('pd', '<u8'), ('Ci', '<i4'), ('Co', '<i4'), ('Cip', '<i4'), ('Cop', '<i4'),
                         ('block_start', '<i4'), ('pad_', '<i4'), ('tapf', '<i4', (9,)), ('tapd', '<i4', (9,))])

    def segnb_pack_pair_job_bytes(self):
        return self.PAIR_JOB.itemsize

    def segnb_pack_pair_job_blocks(self, Co, Ci, Cop, Cip):
        if Co <= 0 or Ci <= 0 or Cop < Co or Cip < Ci or Cop % 8 or Cip % 8:
            return -1
        return ((Cop + 31) // 32) * ((Cip + 63) // 64)

    def segnb_pack_weight_pair_multi(self, jobs, njobs, total_blocks, stream):
        raw = bytes((ctypes.c_char * (njobs * self.PAIR_JOB.itemsize)).from_address(int(jobs)))
        for j in np.frombuffer(raw, dtype=self.PAIR_JOB):
            Ci, Co, Cip, Cop = int(j['Ci']), int(j['Co']), int(j['Cip']), int(j['Cop'])
            W = _mem(int(j['w']), Co * Ci * 9, torch.float32).view(Co, Ci, 9)
            F = _mem(int(j['pf']), Cop * 9 * Cip, torch.bfloat16).view(Cop, 9, Cip)
            F.zero_()
            tf = torch.tensor([int(v) for v in j['tapf']]), torch.tensor([int(v) for v in j['tapd']])
            F[:Co, :, :Ci] = W[:, :, tf[0]].permute(0, 2, 1).to(torch.bfloat16)
            if int(j['pd']):                # (NULL: a layer without a data gradient)
                D = _mem(int(j['pd']), Cip * 9 * Cop, torch.bfloat16).view(Cip, 9, Cop)
                D.zero_()
                D[:Ci, :, :Co] = W[:, :, tf[1]].permute(1, 2, 0).to(torch.bfloat16)
        return 0

    def segnb_sgd_pack_pair_multi(self, jobs, njobs, total_blocks, flat_p, flat_g, lr, stream):
        raw = bytes((ctypes.c_char * (njobs * self.PAIR_JOB.itemsize)).from_address(int(jobs)))
        for j in np.frombuffer(raw, dtype=self.PAIR_JOB):
            n = int(j['Co']) * int(j['Ci']) * 9
            off = (int(j['w']) - int(flat_p)) // 4
            W, G = _mem(int(j['w']), n, torch.float32), _mem(int(flat_g) + 4 * off, n, torch.float32)
            W.sub_(lr * G)
        return self.segnb_pack_weight_pair_multi(jobs, njobs, total_blocks, stream)

    def segnb_sgd_ranges(self, p, g, ranges, nranges, total, lr, stream):
        R = _mem(ranges, 3 * nranges, torch.int64).view(nranges, 3)
        for start, n, _ in R.tolist():
            _mem(int(p) + 4 * start, n, torch.float32).sub_(lr * _mem(int(g) + 4 * start, n, torch.float32))
        return 0

    # element-wise batched forms (jobs segnb_pack_job_blocks refuses): the single-job calls, job by job
    def segnb_pack_elem_job_blocks(self, Mp, Cp, ntaps):
        if Mp <= 0 or Cp <= 0 or ntaps < 1 or ntaps > 64:
            return -1
        return (Mp * ntaps * Cp + 1023) // 1024

    def segnb_pack_weight_elem_multi(self, jobs, njobs, total_blocks, stream):
        for j in self._jobs(jobs, njobs):
            assert not int(j['masked'])
            self.segnb_pack_weight(int(j['w']), int(j['packed']), int(j['dtype']), int(j['Mp']), int(j['Cp']),
                                   int(j['ntaps']), int(j['s_m']), int(j['s_c']),
                                   [int(v) for v in j['tap_off'][:int(j['ntaps'])]], int(j['mmap']), int(j['cmap']), stream)
        return 0

    def segnb_unpack_wgrad_elem_multi(self, jobs, njobs, total_blocks, stream):
        for j in self._jobs(jobs, njobs):
            assert not int(j['masked']) and int(j['nslab']) <= 1
            self.segnb_unpack_wgrad(int(j['packed']), int(j['w']), int(j['Mp']), int(j['Cp']), int(j['ntaps']),
                                    int(j['s_m']), int(j['s_c']), [int(v) for v in j['tap_off'][:int(j['ntaps'])]],
                                    int(j['mmap']), int(j['cmap']), 1, stream)
        return 0

    def segnb_pack_input_nchw(self, x, N, C, H, W, out, dtype, Cp, ld_out, stream):
        X = _mem(x, N * C * H * W, torch.float32).view(N, C, H, W)
        O = _nhwc(out, N, H, W, Cp, ld_out, _tdt(dtype))
        O.zero_()
        O[..., :C] = X.permute(0, 2, 3, 1).to(_tdt(dtype))
        return 0

    # ------------------------------------------------------------------------------------------ BN
    def segnb_bn_finalize(self, stats, C, Cp, count, gamma, beta, eps, momentum, rm, rv, nbt, training, coef, stream):
        co = _mem(coef, 4 * Cp, torch.float32).view(4, Cp)
        co.zero_()
        if training:
            SR = _mem(stats, REPL * 2 * Cp, torch.float64).view(REPL, 2, Cp)
            S = SR.sum(0)
            mu = S[0, :C] / count
            var = (S[1, :C] / count - mu * mu).clamp(min=0)
            if rm is not None:
                RM, RV = _mem(rm, C, torch.float32), _mem(rv, C, torch.float32)
                unb = var * count / (count - 1.0) if count > 1 else var
                RM.copy_(((1 - momentum) * RM.double() + momentum * mu).float())
                RV.copy_(((1 - momentum) * RV.double() + momentum * unb).float())
            if nbt is not None:
                _mem(nbt, 1, torch.int64).add_(1)
            SR.zero_()
        else:
            mu = _mem(rm, C, torch.float32).double()
            var = _mem(rv, C, torch.float32).double()
        invstd = (1.0 / torch.sqrt(var + eps)).float()
        mean = mu.float()
        g = _mem(gamma, C, torch.float32) if gamma is not None else torch.ones(C)
        b = _mem(beta, C, torch.float32) if beta is not None else torch.zeros(C)
        scale = g * invstd
        co[0, :C], co[1, :C], co[2, :C], co[3, :C] = scale, b, mean, invstd
        return 0

    def segnb_bn_finalize_keep(self, stats, C, Cp, count, gamma, beta, eps, momentum, rm, rv, nbt, coef, clear_sums, stream):
        """segnb_bn_finalize(training = 1) that leaves the statistics in place and clears the backward accumulators"""
        keep = _mem(stats, REPL * 2 * Cp, torch.float64).clone()
        rc = self.segnb_bn_finalize(stats, C, Cp, count, gamma, beta, eps, momentum, rm, rv, nbt, 1, coef, stream)
        _mem(stats, REPL * 2 * Cp, torch.float64).copy_(keep)
        if clear_sums is not None:
            _mem(clear_sums, REPL * 2 * Cp, torch.float64).zero_()
        return rc

    @staticmethod
    def _act(z, act, slope):
        if act == ACT_RELU:
            return torch.relu(z)
        if act == ACT_LEAKY:
            return torch.where(z > 0, z, z * slope)
        return z

    @staticmethod
    def _actgrad(z, act, slope):
        if act == ACT_RELU:
            return (z > 0).float()
        if act == ACT_LEAKY:
            return torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
        return torch.ones_like(z)

    def _activated(self, Y, Cp, coef, act, slope, dropmul, N, dt, res=None):
        z = Y.float()
        if coef is not None:
            co = _mem(coef, 4 * Cp, torch.float32).view(4, Cp)
            z = (z - co[2]) * co[0] + co[1]
        if res is not None:
            z = z + res.float()
        a = self._act(z, act, slope)
        dm = None
        if dropmul is not None:
            dm = _mem(dropmul, N * Cp, torch.float32).view(N, 1, 1, Cp)
            a = a * dm
        return z, a.to(dt), dm

    def segnb_bn_act_fwd(self, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, pool_out,
                         ld_pool, up_out, ld_up, res, ld_res, stream):
        dt = _tdt(dtype)
        Y = _nhwc(y, N, H, W, Cp, ld_y, dt)
        R = _nhwc(res, N, H, W, Cp, ld_res, dt) if res is not None else None
        _, a, _ = self._activated(Y, Cp, coef, act, slope, dropmul, N, dt, R)
        if out is not None:
            _nhwc(out, N, H, W, Cp, ld_out, dt).copy_(a)
        if pool_out is not None:
            Hp, Wp = H // 2, W // 2
            p = a[:, :2 * Hp, :2 * Wp].float().reshape(N, Hp, 2, Wp, 2, Cp).amax(dim=(2, 4))
            _nhwc(pool_out, N, Hp, Wp, Cp, ld_pool, dt).copy_(p.to(dt))
        if up_out is not None:
            U = _nhwc(up_out, N, 2 * H, 2 * W, Cp, ld_up, dt)
            U.copy_(a.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2))
        return 0

    def segnb_bn_act_bwd_reduce(self, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, g_direct, ld_gd,
                                g_pool, ld_gp, g_up, ld_gu, dz, ld_dz, sums, res, ld_res, stream):
        dt = _tdt(dtype)
        Y = _nhwc(y, N, H, W, Cp, ld_y, dt)
        R = _nhwc(res, N, H, W, Cp, ld_res, dt) if res is not None else None
        z, a, dm = self._activated(Y, Cp, coef, act, slope, dropmul, N, dt, R)
        g = torch.zeros(N, H, W, Cp)
        if g_direct is not None:
            g += _nhwc(g_direct, N, H, W, Cp, ld_gd, dt).float()
        if g_pool is not None:
            Hp, Wp = H // 2, W // 2
            GP = _nhwc(g_pool, N, Hp, Wp, Cp, ld_gp, dt).float()
            win = a[:, :2 * Hp, :2 * Wp].float().reshape(N, Hp, 2, Wp, 2, Cp).permute(0, 1, 3, 5, 2, 4)
            win = win.reshape(N, Hp, Wp, Cp, 4)
            am = win.argmax(dim=-1)          # torch argmax returns the FIRST maximal index
            onehot = torch.nn.functional.one_hot(am, 4).float() * GP[..., None]
            onehot = onehot.reshape(N, Hp, Wp, Cp, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2 * Hp, 2 * Wp, Cp)
            g[:, :2 * Hp, :2 * Wp] += onehot
        if g_up is not None:
            GU = _nhwc(g_up, N, 2 * H, 2 * W, Cp, ld_gu, dt).float()
            # the kernel adds the four taps in this order
            g = g + GU[:, 0::2, 0::2]
            g = g + GU[:, 0::2, 1::2]
            g = g + GU[:, 1::2, 0::2]
            g = g + GU[:, 1::2, 1::2]
        d = g * self._actgrad(z, act, slope)
        if dm is not None:
            d = g * dm * self._actgrad(z, act, slope)
        d = d.to(dt)
        if dz is not None:
            _nhwc(dz, N, H, W, Cp, ld_dz, dt).copy_(d)
        if sums is not None:
            S = _mem(sums, REPL * 2 * Cp, torch.float64).view(REPL, 2, Cp)[0]
            dd = d.double().reshape(-1, Cp)
            S[0] += dd.sum(0)
            if coef is not None:
                co = _mem(coef, 4 * Cp, torch.float32).view(4, Cp)
                yh = ((Y.float() - co[2]) * co[3]).double().reshape(-1, Cp)
                S[1] += (dd * yh).sum(0)
        return 0

    def segnb_bn_act_bwd_reduce_add(self, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, g_direct, ld_gd, g_add, ld_ga,
                                    dz, ld_dz, sums, res, ld_res, stream):
        dt = _tdt(dtype)
        g = (_nhwc(g_direct, N, H, W, Cp, ld_gd, dt).float() + _nhwc(g_add, N, H, W, Cp, ld_ga, dt).float()).to(dt).contiguous()
        return self.segnb_bn_act_bwd_reduce(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, g.data_ptr(), Cp, None, 0,
                                            None, 0, dz, ld_dz, sums, res, ld_res, stream)

    def segnb_bn_bwd_finalize(self, sums, C, Cp, count, gamma, coef, bcoef, dgamma, dbeta, accumulate, stream):
        SR = _mem(sums, REPL * 2 * Cp, torch.float64).view(REPL, 2, Cp)
        S = SR.sum(0)
        co = _mem(coef, 4 * Cp, torch.float32).view(4, Cp)
        bc = _mem(bcoef, 3 * Cp, torch.float32).view(3, Cp)
        bc.zero_()
        g = _mem(gamma, C, torch.float32) if gamma is not None else torch.ones(C)
        bc[0, :C] = g * co[3, :C]
        bc[1, :C] = (S[0, :C] / count).float()
        bc[2, :C] = (S[1, :C] / count).float()
        if dgamma is not None:
            G = _mem(dgamma, C, torch.float32)
            G.copy_((G if accumulate else 0) + S[1, :C].float())
        if dbeta is not None:
            B = _mem(dbeta, C, torch.float32)
            B.copy_((B if accumulate else 0) + S[0, :C].float())
        SR.zero_()
        return 0

    def segnb_bn_bwd_finalize_clear(self, sums, C, Cp, count, gamma, coef, bcoef, dgamma, dbeta, accumulate, clear, stream):
        rc = self.segnb_bn_bwd_finalize(sums, C, Cp, count, gamma, coef, bcoef, dgamma, dbeta, accumulate, stream)
        if clear is not None:
            _mem(clear, REPL * 2 * Cp, torch.float64).zero_()
        return rc

    def segnb_bn_bwd_apply(self, dtype, y, ld_y, N, H, W, Cp, coef, bcoef, dz, ld_dz, dy, ld_dy, dbias, C, stream):
        dt = _tdt(dtype)
        Y = _nhwc(y, N, H, W, Cp, ld_y, dt).float()
        D = _nhwc(dz, N, H, W, Cp, ld_dz, dt).float()
        co = _mem(coef, 4 * Cp, torch.float32).view(4, Cp)
        bc = _mem(bcoef, 3 * Cp, torch.float32).view(3, Cp)
        yh = (Y - co[2]) * co[3]
        out = (bc[0] * (D - bc[1] - yh * bc[2])).to(dt)
        _nhwc(dy, N, H, W, Cp, ld_dy, dt).copy_(out)
        if dbias is not None:
            _mem(dbias, C, torch.float32).add_(out.float().reshape(-1, Cp).sum(0)[:C])
        return 0

    def segnb_bn_bwd_apply_direct(self, dtype, y, ld_y, N, H, W, Cp, coef, bcoef, act, slope, g, ld_g, dy, ld_dy,
                                  dbias, C, stream):
        """segnb_bn_act_bwd_reduce's dz (never stored) recomputed from g, then segnb_bn_bwd_apply"""
        dt = _tdt(dtype)
        Y = _nhwc(y, N, H, W, Cp, ld_y, dt)
        z, _, _ = self._activated(Y, Cp, coef, act, slope, None, N, dt, None)
        D = (_nhwc(g, N, H, W, Cp, ld_g, dt).float() * self._actgrad(z, act, slope)).to(dt).float()
        co = _mem(coef, 4 * Cp, torch.float32).view(4, Cp)
        bc = _mem(bcoef, 3 * Cp, torch.float32).view(3, Cp)
        yh = (Y.float() - co[2]) * co[3]
        out = (bc[0] * (D - bc[1] - yh * bc[2])).to(dt)
        _nhwc(dy, N, H, W, Cp, ld_dy, dt).copy_(out)
        if dbias is not None:
            _mem(dbias, C, torch.float32).add_(out.float().reshape(-1, Cp).sum(0)[:C])
        return 0

    # ------------------------------------------------------------------------------------------ generic NHWC ops
    def segnb_bn_fwd_fused(self, dtype, y, ld_y, N, H, W, C, Cp, stats, gamma, beta, eps, momentum, rm, rv, nbt, coef,
                           clear_sums, act, slope, dropmul, out, ld_out, pool_out, ld_pool, up_out, ld_up, res, ld_res,
                           stream):
        """finalize + act_fwd; the forward statistics are left as they are, the backward sums are cleared."""
        keep = _mem(stats, REPL * 2 * Cp, torch.float64).clone()
        rc = self.segnb_bn_finalize(stats, C, Cp, float(N * H * W), gamma, beta, eps, momentum, rm, rv, nbt, 1, coef,
                                    stream)
        _mem(stats, REPL * 2 * Cp, torch.float64).copy_(keep)
        if clear_sums is not None:
            _mem(clear_sums, REPL * 2 * Cp, torch.float64).zero_()
        return rc or self.segnb_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out,
                                           pool_out, ld_pool, up_out, ld_up, res, ld_res, stream)

    def segnb_bn_bwd_apply_fused(self, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                 accumulate, clear_stats, dz, ld_dz, dy, ld_dy, stream):
        keep = _mem(sums, REPL * 2 * Cp, torch.float64).clone()
        rc = self.segnb_bn_bwd_finalize(sums, C, Cp, float(N * H * W), gamma, coef, bcoef, dgamma, dbeta, accumulate,
                                        stream)
        _mem(sums, REPL * 2 * Cp, torch.float64).copy_(keep)
        if clear_stats is not None:
            _mem(clear_stats, REPL * 2 * Cp, torch.float64).zero_()
        return rc or self.segnb_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, bcoef, dz, ld_dz, dy, ld_dy, None, C,
                                             stream)

    def segnb_bn_bwd_apply_fused_acc(self, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                     accumulate, clear_stats, dz, ld_dz, dy, ld_dy, stream):
        dt = _tdt(dtype)
        old = _nhwc(dy, N, H, W, Cp, ld_dy, dt).clone()
        rc = self.segnb_bn_bwd_apply_fused(dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                           accumulate, clear_stats, dz, ld_dz, dy, ld_dy, stream)
        o = _nhwc(dy, N, H, W, Cp, ld_dy, dt)
        o.copy_((old.float() + o.float()).to(dt))
        return rc

    def segnb_bn_bwd_apply_fused_direct(self, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                        accumulate, clear_stats, act, slope, g, ld_g, dy, ld_dy, stream):
        keep = _mem(sums, REPL * 2 * Cp, torch.float64).clone()
        rc = self.segnb_bn_bwd_finalize(sums, C, Cp, float(N * H * W), gamma, coef, bcoef, dgamma, dbeta, accumulate,
                                        stream)
        _mem(sums, REPL * 2 * Cp, torch.float64).copy_(keep)
        if clear_stats is not None:
            _mem(clear_stats, REPL * 2 * Cp, torch.float64).zero_()
        return rc or self.segnb_bn_bwd_apply_direct(dtype, y, ld_y, N, H, W, Cp, coef, bcoef, act, slope, g, ld_g, dy,
                                                    ld_dy, None, C, stream)

    def segnb_bn_bwd_apply_fused_direct_acc(self, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                            accumulate, clear_stats, act, slope, g, ld_g, dy, ld_dy, stream):
        dt = _tdt(dtype)
        old = _nhwc(dy, N, H, W, Cp, ld_dy, dt).clone()
        rc = self.segnb_bn_bwd_apply_fused_direct(dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                                  accumulate, clear_stats, act, slope, g, ld_g, dy, ld_dy, stream)
        o = _nhwc(dy, N, H, W, Cp, ld_dy, dt)
        o.copy_((old.float() + o.float()).to(dt))
        return rc

    # ------------------------------------------------------------------------------------------ tiles
    def segnb_bn_bwd_apply_fused_src(self, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                     accumulate, clear_stats, act, slope, dropmul, g_direct, ld_gd, g_pool, ld_gp, g_up, ld_gu,
                                     dy, ld_dy, stream):
        """dz recomputed from the sources (segnb_bn_act_bwd_reduce into a temporary, no sums), then segnb_bn_bwd_apply_fused"""
        tmp = torch.zeros(N * H * W * Cp, dtype=_tdt(dtype))
        rc = self.segnb_bn_act_bwd_reduce(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, g_direct, ld_gd, g_pool,
                                          ld_gp, g_up, ld_gu, tmp.data_ptr(), Cp, None, None, 0, stream)
        if rc:
            return rc
        return self.segnb_bn_bwd_apply_fused(dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta,
                                             accumulate, clear_stats, tmp.data_ptr(), Cp, dy, ld_dy, stream)

    def segnb_abn_scale(self, w, eps, out, n, stream):
        _mem(out, n, torch.float32).copy_(_mem(w, n, torch.float32).abs() + eps)
        return 0

    def segnb_abn_dscale(self, w, dscale, dw, n, stream):
        W, D = _mem(w, n, torch.float32), _mem(dscale, n, torch.float32)
        _mem(dw, n, torch.float32).add_(torch.where(W > 0, D, -D))
        D.zero_()
        return 0

    @staticmethod
    def _d4(k, t):
        """element k of tta_d4_aug applied to an [..., S, S] tensor (rot90 counter-clockwise, then fliplr)"""
        t = torch.rot90(t, k & 3, dims=(-2, -1))
        return torch.flip(t, dims=(-1,)) if k >= 4 else t

    def segnb_tiles_gather(self, image, H, W, C, mt, ml, crops, first, count, S, out, stream):
        img = _mem(image, H * W * C, torch.float32).view(H, W, C)
        cr = _mem(crops, 2 * ((first + count + 7) // 8), torch.int32).view(-1, 2)
        O = _mem(out, count * C * S * S, torch.float32).view(count, C, S, S)
        ys, xs = torch.arange(S), torch.arange(S)
        for b in range(count):
            tile, k = (first + b) >> 3, (first + b) & 7
            yy = (int(cr[tile, 1]) + ys - mt).abs()
            yy = torch.where(yy >= H, 2 * H - 2 - yy, yy)
            xx = (int(cr[tile, 0]) + xs - ml).abs()
            xx = torch.where(xx >= W, 2 * W - 2 - xx, xx)
            t = img[yy][:, xx].permute(2, 0, 1)                       # [C, S, S]
            O[b] = self._d4(k, t)
        return 0

    def segnb_tiles_gather_u8(self, image, H, W, C, mt, ml, crops, first, count, S, scale, mean, stdv, out, stream):
        # NormalizeImage of lib/augmentations.py:452-460 on the uint8 image, then the float gather
        img = _mem(image, H * W * C, torch.uint8).view(H, W, C).float()
        mean_t = torch.tensor([float(mean[c]) for c in range(C)])
        istd_t = torch.tensor([1.0 / float(stdv[c]) for c in range(C)], dtype=torch.float32)
        norm = ((img * float(scale) - mean_t) * istd_t).contiguous()
        return self.segnb_tiles_gather(norm.data_ptr(), H, W, C, mt, ml, crops, first, count, S, out, stream)

    def segnb_tiles_merge(self, logits, K, S, crops, ntiles, step, nx, ny, weight, H, W, mt, ml, out, stream):
        L = _mem(logits, ntiles * 8 * K * S * S, torch.float32).view(ntiles, 8, K, S, S)
        cr = _mem(crops, 2 * ntiles, torch.int32).view(ntiles, 2)
        Wt = _mem(weight, S * S, torch.float64).view(S, S)
        Hp = int(cr[:, 1].max()) + S
        Wp = int(cr[:, 0].max()) + S
        acc = torch.zeros(K, Hp, Wp, dtype=torch.float64)
        norm = torch.zeros(Hp, Wp, dtype=torch.float64)
        for t in range(ntiles):
            p = torch.sigmoid(L[t])
            s = torch.zeros(K, S, S)
            for k in range(8):                                        # undo transform k: fliplr first, then rot back
                u = torch.flip(p[k], dims=(-1,)) if k >= 4 else p[k]
                s = s + torch.rot90(u, -(k & 3), dims=(-2, -1))
            v = s * 0.125
            x, y = int(cr[t, 0]), int(cr[t, 1])
            acc[:, y:y + S, x:x + S] += v.double() * Wt
            norm[y:y + S, x:x + S] += Wt
        res = (acc / norm.clamp(min=2.220446049250313e-16)).float()
        _mem(out, H * W * K, torch.float32).view(H, W, K).copy_(res[:, mt:mt + H, ml:ml + W].permute(1, 2, 0))
        return 0

    def segnb_add(self, dtype, a, ld_a, b, ld_b, out, ld_out, N, H, W, Cp, stream):
        dt = _tdt(dtype)
        o = _nhwc(out, N, H, W, Cp, ld_out, dt)
        r = torch.zeros(o.shape, dtype=torch.float32)
        for src, ld in ((a, ld_a), (b, ld_b)):          # a NULL operand counts as zeros
            if src:
                r = r + _nhwc(src, N, H, W, Cp, ld, dt).float()
        o.copy_(r.to(dt))
        return 0

    def segnb_bn_stats(self, dtype, x, ld, N, H, W, Cp, stats, stream):
        v = _nhwc(x, N, H, W, Cp, ld, _tdt(dtype)).double().reshape(-1, Cp)
        S = _mem(stats, REPL * 2 * Cp, torch.float64).view(REPL, 2, Cp)[0]
        S[0] += v.sum(0)
        S[1] += (v * v).sum(0)
        return 0

    # ---- statistics as a channel range of a wider [REPL][2][ld] table (FCDenseNet's concat prefixes)
    @staticmethod
    def _stats_range(stats, Cp, ld):
        """[REPL][2][Cp] strided view of the range"""
        flat = _mem(stats, (REPL * 2 - 1) * ld + Cp, torch.float64)
        return flat.as_strided((REPL, 2, Cp), (2 * ld, ld, 1))

    def segnb_bn_stats_ld(self, dtype, x, ld, N, H, W, Cp, stats, stats_ld, stream):
        v = _nhwc(x, N, H, W, Cp, ld, _tdt(dtype)).double().reshape(-1, Cp)
        S = self._stats_range(stats, Cp, stats_ld)[0]
        S[0] += v.sum(0)
        S[1] += (v * v).sum(0)
        return 0

    def segnb_bn_act_fwd_stats(self, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, out_stats,
                               out_stats_ld, stream):
        rc = self.segnb_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, None, 0, None, 0,
                                   None, 0, stream)
        return rc or self.segnb_bn_stats_ld(dtype, out, ld_out, N, H, W, Cp, out_stats, out_stats_ld, stream)

    def segnb_bn_fwd_fused_ld(self, dtype, y, ld_y, N, H, W, C, Cp, stats, stats_ld, gamma, beta, eps, momentum, rm, rv, nbt,
                              coef, clear_sums, act, slope, dropmul, out, ld_out, stream):
        dense = self._stats_range(stats, Cp, stats_ld).contiguous().clone()       # the range as a [REPL][2][Cp] table of its own
        return self.segnb_bn_fwd_fused(dtype, y, ld_y, N, H, W, C, Cp, dense.data_ptr(), gamma, beta, eps, momentum, rm, rv,
                                       nbt, coef, clear_sums, act, slope, dropmul, out, ld_out, None, 0, None, 0, None, 0,
                                       stream)

    def segnb_upsample_bilinear2x_fwd(self, dtype, x, ld_x, N, H, W, Cp, out, ld_out, stream):
        # nn.Upsample(scale_factor=2, mode='bilinear') of lib/models/unet16.py:43 (align_corners=False)
        dt = _tdt(dtype)
        X = _nhwc(x, N, H, W, Cp, ld_x, dt).float().permute(0, 3, 1, 2)
        U = torch.nn.functional.interpolate(X, scale_factor=2, mode='bilinear', align_corners=False)
        _nhwc(out, N, 2 * H, 2 * W, Cp, ld_out, dt).copy_(U.permute(0, 2, 3, 1).to(dt))
        return 0

    def segnb_upsample_bilinear2x_bwd(self, dtype, g_out, ld_go, N, H, W, Cp, dx, ld_dx, stream):
        dt = _tdt(dtype)
        with torch.enable_grad():
            X = torch.zeros(N, Cp, H, W, requires_grad=True)
            U = torch.nn.functional.interpolate(X, scale_factor=2, mode='bilinear', align_corners=False)
            G = _nhwc(g_out, N, 2 * H, 2 * W, Cp, ld_go, dt).float().permute(0, 3, 1, 2)
            U.backward(G)
        _nhwc(dx, N, H, W, Cp, ld_dx, dt).copy_(X.grad.permute(0, 2, 3, 1).to(dt))
        return 0

    def segnb_maxpool_fwd(self, dtype, x, ld_x, N, H, W, Cp, k, stride, pad, out, ld_out, idx, stream):
        dt = _tdt(dtype)
        X = _nhwc(x, N, H, W, Cp, ld_x, dt).float().permute(0, 3, 1, 2)
        P, I = torch.nn.functional.max_pool2d(X, k, stride, pad, return_indices=True)
        _nhwc(out, N, P.shape[2], P.shape[3], Cp, ld_out, dt).copy_(P.permute(0, 2, 3, 1).to(dt))
        if idx is not None:               # flat input index -> window position a*k + b
            Ho, Wo = P.shape[2], P.shape[3]
            hi, wi = I // W, I % W
            a = hi - (torch.arange(Ho)[:, None] * stride - pad)
            b = wi - (torch.arange(Wo)[None, :] * stride - pad)
            _mem(idx, N * Ho * Wo * Cp, torch.uint8).view(N, Ho, Wo, Cp).copy_((a * k + b).permute(0, 2, 3, 1).to(torch.uint8))
        return 0

    def segnb_maxpool_bwd(self, dtype, x, ld_x, g_out, ld_go, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, stream):
        dt = _tdt(dtype)
        with torch.enable_grad():      # called from inside autograd.Function.backward (grad mode off there)
            X = _nhwc(x, N, H, W, Cp, ld_x, dt).float().permute(0, 3, 1, 2).clone().requires_grad_(True)
            P = torch.nn.functional.max_pool2d(X, k, stride, pad)
            G = _nhwc(g_out, N, P.shape[2], P.shape[3], Cp, ld_go, dt).float().permute(0, 3, 1, 2)
            P.backward(G)
        _nhwc(dx, N, H, W, Cp, ld_dx, dt).copy_(X.grad.permute(0, 2, 3, 1).to(dt))
        return 0

    def segnb_maxpool_bwd_add(self, dtype, x, ld_x, g_out, ld_go, g_out2, ld_go2, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, stream):
        dt = _tdt(dtype)
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        g = (_nhwc(g_out, N, Ho, Wo, Cp, ld_go, dt).float() + _nhwc(g_out2, N, Ho, Wo, Cp, ld_go2, dt).float()).to(dt).contiguous()
        return self.segnb_maxpool_bwd(dtype, x, ld_x, g.data_ptr(), Cp, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, stream)

    def segnb_nhwc_to_nchw_f32(self, dtype, a, ld, N, H, W, C, out, stream):
        A = _nhwc(a, N, H, W, C, ld, _tdt(dtype)).float()
        _mem(out, N * C * H * W, torch.float32).view(N, C, H, W).copy_(A.permute(0, 3, 1, 2))
        return 0

    # ------------------------------------------------------------------------------------------ head
    # ---- the last BatchNorm + activation layer and the classifier behind it as one pass (the two-launch forms, composed)
    def segnb_bias_grad_job_bytes(self):
        return 24

    def segnb_bias_grad_multi(self, jobs, njobs, stream):
        import ctypes
        raw = (ctypes.c_uint8 * (24 * njobs)).from_address(jobs)
        tab = np.frombuffer(bytes(raw), dtype=np.dtype([('sums', '<u8'), ('gb', '<u8'), ('C', '<i4'), ('Cp', '<i4')]))
        for j in tab:
            C, Cp = int(j['C']), int(j['Cp'])
            S = _mem(int(j['sums']), REPL * 2 * Cp, torch.float64).view(REPL, 2, Cp)
            if int(j['gb']):
                _mem(int(j['gb']), C, torch.float32).add_(S[:, 0, :C].sum(0).float())
            S.zero_()
        return 0

    def segnb_head_fused_ok(self, K, Cp):
        cpp = Cp // 8
        return int(1 <= K <= 4 and Cp % 8 == 0 and 8 <= Cp <= 256 and cpp & (cpp - 1) == 0)

    def segnb_bn_fwd_fused_head(self, dtype, y, ld_y, N, H, W, C, Cp, stats, gamma, beta, eps, momentum, rm, rv, nbt, coef,
                                clear_sums, act, slope, dropmul, out, ld_out, head_w, head_b, K, logits, stream):
        if not self.segnb_head_fused_ok(K, Cp):
            return -1
        tmp = torch.zeros(N * H * W * Cp, dtype=_tdt(dtype))
        rc = self.segnb_bn_fwd_fused(dtype, y, ld_y, N, H, W, C, Cp, stats, gamma, beta, eps, momentum, rm, rv, nbt, coef,
                                     clear_sums, act, slope, dropmul, tmp.data_ptr(), Cp, None, 0, None, 0, None, 0, stream)
        if rc:
            return rc
        if out is not None:
            _nhwc(out, N, H, W, Cp, ld_out, _tdt(dtype)).copy_(tmp.view(N, H, W, Cp))
        return self.segnb_head_fwd(dtype, tmp.data_ptr(), Cp, N, H, W, C, head_w, head_b, K, logits, stream)

    def segnb_head_bn_bwd(self, dtype, y, ld_y, N, H, W, C, Cp, coef, act, slope, dropmul, head_w, K, dlogits, dz, ld_dz, sums,
                          dw, db, stream):
        if not self.segnb_head_fused_ok(K, Cp):
            return -1
        dt = _tdt(dtype)
        a = torch.zeros(N * H * W * Cp, dtype=dt)            # the activated tensor, recomputed
        rc = self.segnb_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, a.data_ptr(), Cp, None, 0, None, 0,
                                   None, 0, stream)
        if rc:
            return rc
        da = torch.zeros(N * H * W * Cp, dtype=dt)
        rc = self.segnb_head_bwd(dtype, a.data_ptr(), Cp, N, H, W, C, Cp, head_w, K, dlogits, da.data_ptr(), Cp, dw, db, stream)
        if rc:
            return rc
        return self.segnb_bn_act_bwd_reduce(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, da.data_ptr(), Cp, None, 0,
                                            None, 0, dz, ld_dz, sums, None, 0, stream)

    def segnb_head_bn_bwd_apply(self, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate,
                                clear_stats, act, slope, dropmul, head_w, K, dlogits, dy, ld_dy, stream):
        """dz recomputed as segnb_head_bn_bwd computes it (into a temporary), then segnb_bn_bwd_apply_fused on it"""
        if not self.segnb_head_fused_ok(K, Cp):
            return -1
        dt = _tdt(dtype)
        da = torch.zeros(N * H * W * Cp, dtype=dt)
        rc = self.segnb_head_bwd(dtype, None, Cp, N, H, W, C, Cp, head_w, K, dlogits, da.data_ptr(), Cp, None, None, stream)
        if rc:
            return rc
        dz = torch.zeros(N * H * W * Cp, dtype=dt)
        rc = self.segnb_bn_act_bwd_reduce(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, da.data_ptr(), Cp, None, 0,
                                          None, 0, dz.data_ptr(), Cp, None, None, 0, stream)
        if rc:
            return rc
        return self.segnb_bn_bwd_apply_fused(dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate,
                                             clear_stats, dz.data_ptr(), Cp, dy, ld_dy, stream)

    def segnb_head_fwd(self, dtype, a, ld_a, N, H, W, C, w, bias, K, logits, stream):
        A = _nhwc(a, N, H, W, C, ld_a, _tdt(dtype)).float()
        Wm = _mem(w, K * C, torch.float32).view(K, C)
        out = A @ Wm.t()
        if bias is not None:
            out = out + _mem(bias, K, torch.float32)
        _mem(logits, N * K * H * W, torch.float32).view(N, K, H, W).copy_(out.permute(0, 3, 1, 2))
        return 0

    def segnb_head_bwd(self, dtype, a, ld_a, N, H, W, C, Cp, w, K, dlogits, da, ld_da, dw, db, stream):
        dt = _tdt(dtype)
        A = _nhwc(a, N, H, W, C, ld_a, dt).float() if a is not None else None
        Wm = _mem(w, K * C, torch.float32).view(K, C)
        DL = _mem(dlogits, N * K * H * W, torch.float32).view(N, K, H, W).permute(0, 2, 3, 1)
        if da is not None:
            DA = _nhwc(da, N, H, W, Cp, ld_da, dt)
            DA.zero_()
            DA[..., :C] = (DL @ Wm).to(dt)
        if dw is not None:
            _mem(dw, K * C, torch.float32).view(K, C).add_(DL.reshape(-1, K).t() @ A.reshape(-1, C))
        if db is not None:
            _mem(db, K, torch.float32).add_(DL.reshape(-1, K).sum(0))
        return 0

    def segnb_head_conv_ok(self, C, K, kh, kw):
        return int(K >= 1 and kh >= 1 and kw >= 1 and K * kh * kw <= 8 and 1 <= C <= 64)

    def segnb_head_conv_fwd(self, dtype, a, ld_a, N, Hi, Wi, C, w, kh, kw, pad, bias, K, logits, stream):
        A = _nhwc(a, N, Hi, Wi, C, ld_a, _tdt(dtype)).float().permute(0, 3, 1, 2)
        Wm = _mem(w, K * C * kh * kw, torch.float32).view(K, C, kh, kw)
        b = _mem(bias, K, torch.float32) if bias is not None else None
        out = torch.nn.functional.conv2d(A, Wm, b, padding=pad)
        _mem(logits, out.numel(), torch.float32).view(out.shape).copy_(out)
        return 0

    def segnb_head_conv_bwd(self, dtype, a, ld_a, N, Hi, Wi, C, Cp, w, kh, kw, pad, K, dlogits, act, slope, da, ld_da, dw, db,
                            sums, stream):
        dt = _tdt(dtype)
        Av = _nhwc(a, N, Hi, Wi, Cp, ld_a, dt)
        A = Av[..., :C].float().permute(0, 3, 1, 2)
        Wm = _mem(w, K * C * kh * kw, torch.float32).view(K, C, kh, kw)
        Ho, Wo = Hi + 2 * pad - kh + 1, Wi + 2 * pad - kw + 1
        DL = _mem(dlogits, N * K * Ho * Wo, torch.float32).view(N, K, Ho, Wo)
        if da is not None:
            g = torch.nn.grad.conv2d_input((N, C, Hi, Wi), Wm, DL, padding=pad).permute(0, 2, 3, 1)
            DA = _nhwc(da, N, Hi, Wi, Cp, ld_da, dt)
            full = torch.zeros(N, Hi, Wi, Cp)
            full[..., :C] = g
            full = full.to(dt)
            if act >= 0:
                neg = 0.0 if act == ACT_RELU else (slope if act == ACT_LEAKY else 1.0)
                full = torch.where(Av.float() > 0, full.float(), full.float() * neg).to(dt)
                if sums is not None:
                    S = _mem(sums, REPL * 2 * Cp, torch.float64).view(REPL, 2, Cp)[0]
                    S[0] += full.double().reshape(-1, Cp).sum(0)
            DA.copy_(full)
        if dw is not None:
            _mem(dw, K * C * kh * kw, torch.float32).view(K, C, kh, kw).add_(
                torch.nn.grad.conv2d_weight(A, (K, C, kh, kw), DL, padding=pad))
        if db is not None:
            _mem(db, K, torch.float32).add_(DL.sum((0, 2, 3)))
        return 0

    # ------------------------------------------------------------------------------------------ loss
    @staticmethod
    def _terms(x, t, gamma=2.0):
        ls = torch.nn.functional.logsigmoid(x)
        p = torch.exp(ls)
        e = -t * ls + torch.log1p(p)
        de = (1 - p) * (p / (1 + p) - t)
        pt = torch.exp(-e)
        om = 1 - pt
        if gamma == 2.0:
            f = om * om * e
            df = (2 * om * pt * e + om * om) * de
        else:
            pg1 = om.clamp_min(1e-38).pow(gamma - 1.0)
            pg = pg1 * om
            f = pg * e
            df = (gamma * pg1 * pt * e + pg) * de
        return p, e, de, f, df

    def segnb_seg_loss_reduce(self, logits, target, n, focal_gamma, sums, stream):
        x = _mem(logits, n, torch.float32)
        t = _mem(target, n, torch.int64).float()
        p, e, _, f, _ = self._terms(x, t, float(focal_gamma))
        S = _mem(sums, 8, torch.float64)
        S[0] += e.double().sum()
        S[1] += f.double().sum()
        S[2] += (p * t).double().sum()
        S[3] += p.double().sum()
        S[4] += t.double().sum()
        S[5] += ((p > 0.5) == (t != 0)).double().sum()
        S[6] += float(n)
        return 0

    def segnb_seg_loss_reduce_finalize(self, logits, target, n, spec, work, out, stream):
        """reduce + finalize; the work buffer comes in zeroed and leaves zeroed"""
        sp = _geom(spec)
        rc = self.segnb_seg_loss_reduce(logits, target, n, float(sp.focal_gamma), work, stream)
        rc = rc or self.segnb_seg_loss_finalize(work, spec, out, stream)
        _mem(work, 128, torch.float64).zero_()
        return rc

    def segnb_seg_loss_finalize(self, sums, spec, out, stream):
        sp = _geom(spec)
        S = _mem(sums, 8, torch.float64).tolist()
        n, I, U = S[6], S[2], S[3] + S[4]
        bce = S[0] if sp.bce_sum else S[0] / n
        focal = S[1] / n if sp.focal_mean else S[1]
        eps, sm = float(sp.eps), float(sp.smooth)
        Dj, Ds, Dd = U - I + eps, U - I + sm, U + eps
        loss = (sp.w_bce * bce + sp.w_focal * focal + sp.w_jaccard * (1 - I / Dj) +
                sp.w_sjaccard * (1 - (I + sm) / Ds) + sp.w_dice * (1 - 2 * I / Dd)) / sp.norm
        GI = sp.w_jaccard * (-(U + eps) / Dj ** 2) + sp.w_sjaccard * (-(U + 2 * sm) / Ds ** 2) + sp.w_dice * (-2 / Dd)
        GU = sp.w_jaccard * (I / Dj ** 2) + sp.w_sjaccard * ((I + sm) / Ds ** 2) + sp.w_dice * (2 * I / Dd ** 2)
        _mem(out, 8, torch.float32).copy_(torch.tensor([loss, I / (U - I + 1e-7), S[5] / n, GI, GU, bce, n, 0.0]))
        return 0

    def segnb_seg_loss_bwd(self, logits, target, n, sums, fin, spec, grad_out, dlogits, stream):
        sp = _geom(spec)
        x = _mem(logits, n, torch.float32)
        t = _mem(target, n, torch.int64).float()
        F = _mem(fin, 8, torch.float32)
        go = (float(_mem(grad_out, 1, torch.float32)[0]) if grad_out is not None else 1.0) / sp.norm
        p, _, de, _, df = self._terms(x, t, float(sp.focal_gamma))
        inv_n = 1.0 / float(F[6])
        wf = sp.w_focal * inv_n if sp.focal_mean else sp.w_focal
        wb = sp.w_bce if sp.bce_sum else sp.w_bce * inv_n
        dx = go * (wb * de + wf * df + p * (1 - p) * (t * float(F[3]) + float(F[4])))
        _mem(dlogits, n, torch.float32).copy_(dx)
        return 0

    def segnb_seg_loss_map(self, logits, target, n, kind, gamma, out, stream):
        _, e, _, f, _ = self._terms(_mem(logits, n, torch.float32), _mem(target, n, torch.int64).float(), float(gamma))
        _mem(out, n, torch.float32).copy_(e if kind == 0 else f)
        return 0

    def segnb_seg_loss_map_bwd(self, logits, target, n, kind, gamma, grad_out, dlogits, stream):
        _, _, de, _, df = self._terms(_mem(logits, n, torch.float32), _mem(target, n, torch.int64).float(), float(gamma))
        _mem(dlogits, n, torch.float32).copy_(_mem(grad_out, n, torch.float32) * (de if kind == 0 else df))
        return 0

    def segnb_absmax_f32(self, x, n, out, stream):
        _mem(out, 1, torch.float32)[0] = _mem(x, n, torch.float32).abs().max()
        return 0

    def segnb_pr_histogram(self, logits, target, n, thresholds, nthr, hist, stream):
        """train_utils.py:109-125 restated as a histogram: bucket = #thresholds strictly below sigmoid(x)"""
        p = torch.exp(torch.nn.functional.logsigmoid(_mem(logits, n, torch.float32)))
        t = _mem(target, n, torch.int64) != 0
        b = torch.bucketize(p, _mem(thresholds, nthr, torch.float32), right=False)
        H = _mem(hist, 2 * (nthr + 1), torch.int64).view(2, nthr + 1)
        H[0] += torch.bincount(b[~t], minlength=nthr + 1)
        H[1] += torch.bincount(b[t], minlength=nthr + 1)
        return 0

    def segnb_stream_fork(self, main, side):
        return 0

    def segnb_stream_fork_arm(self, main):
        return 0

    def segnb_stream_fork_commit(self, main, side):
        return 0

    def segnb_event_record(self, event, stream):
        return 0

    def segnb_stream_join(self, main, side):
        return 0

    def segnb_wg_cu_share(self, pct):
        return self.segnb_tune('wg_cu_pct', pct)

    def segnb_tune(self, key, value):
        self.tuned = getattr(self, 'tuned', {})
        self.tuned[key.decode() if isinstance(key, bytes) else key] = int(value)
        return 0

    def segnb_device_cus(self):
        return 256

    def segnb_debug_stamps(self, host_dst):
        return 0

    def segnb_debug_census(self, buf, cap):       # (no launch lists on the CPU: nothing to compare)
        return 0

    def segnb_sgd_step(self, p, g, n, lr, stream):
        _mem(p, n, torch.float32).sub_(lr * _mem(g, n, torch.float32))
        return 0

    def segnb_rmsprop_step(self, p, g, sq, n, lr, alpha, eps, stream):
        P, G, V = (_mem(t, n, torch.float32) for t in (p, g, sq))
        V.mul_(alpha).addcmul_(G, G, value=1 - alpha)
        P.addcdiv_(G, V.sqrt().add_(eps), value=-lr)
        return 0

    def segnb_adam_step(self, p, g, m, v, n, lr, beta1, beta2, eps, step, stream):
        P, G, M, V = (_mem(t, n, torch.float32) for t in (p, g, m, v))
        M.lerp_(G, 1 - beta1)
        V.mul_(beta2).addcmul_(G, G, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        P.addcdiv_(M, (V.sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)
        return 0
