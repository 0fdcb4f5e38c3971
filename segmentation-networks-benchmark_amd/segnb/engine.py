"""Host-side execution engine: NHWC activation views, packed-weight convolution ops, fused
conv->BatchNorm->activation stages, flat parameter/gradient storage.

Everything here is plumbing around the C ABI (segnb._native): PyTorch supplies device memory and the
stream; every FLOP and every byte moved on the hot path happens inside libsegnb_hip.so.  The same
code drives any device the ABI backend can address, which is what lets tests check the plan logic
(buffer wiring, tap tables, channel maps, backward routing) on CPU against an ABI emulator.
"""
import os

import numpy as np
import torch

from . import _native as nv
from . import convplan as cp

BN_EPS = 1e-5
STAT_REPLICAS = 16     # SEGNB_STAT_REPLICAS
BN_MOMENTUM = 0.1


class KernelTimer(object):
    """Optional per-launch timing of the convolution kernels with HIP events on the launch stream
    (bench.py's live roofline measurement).  Off unless bench.py installs one in ``engine.TIMER``."""

    def __init__(self):
        self.records = []      # (label, algorithmic_flops, start_event, end_event)
        self.acc = {}
        self.executed, self.algorithmic = {}, {}      # per label, summed over the launches SEEN by launch() (recording steps)
        self.persistent = False    # the events are part of a recorded launch list: re-recorded by every replay

    def launch(self, label, flops, fn, executed=None):
        """flops: ALGORITHMIC count of the launch (the reference's 3x3 taps); executed: what the kernel
        really issues when that differs (sub-pixel form of Upsample -> conv: 4 taps instead of 9)."""
        self.executed[label] = self.executed.get(label, 0.0) + (flops if executed is None else executed)
        self.algorithmic[label] = self.algorithmic.get(label, 0.0) + flops
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()             # (creates the HIP event; recorded again below through the ABI)
        b.record()
        st = torch.cuda.current_stream().cuda_stream
        # through the ABI, so that a launch list being recorded contains the two records around the launch
        nv.call('segnb_event_record', a.cuda_event, st)
        fn()
        nv.call('segnb_event_record', b.cuda_event, st)
        self.records.append((label, flops, a, b))

    def collect(self):
        """After a device synchronize at the end of a step: add the step's launch durations to the totals."""
        for label, flops, a, b in self.records:
            n, ms, fl = self.acc.get(label, (0, 0.0, 0.0))
            self.acc[label] = (n + 1, ms + a.elapsed_time(b), fl + flops)
        if not self.persistent:
            self.records = []

    def summary(self):
        """{label: (launches, total_ms, total_flops)} -- call after a device synchronize."""
        if self.records and not self.persistent:
            self.collect()
        return dict(self.acc)


TIMER = None


def _timed(label, flops, fn, executed=None):
    if TIMER is None:
        fn()
    else:
        TIMER.launch(label, flops, fn, executed)


class ReplayGuard(object):
    """Structural check of record / replay (SEGNB_REPLAY_GUARD=1, or ``ReplayGuard.enable()``; debug mode, GPU only).

    A model's forward / backward is either run by the Python launcher while the library records its ABI calls, or replayed from
    the recorded lists (segnb_plan_run) with whatever host code sits around them.  Nothing ties the two together: a launch that
    the recording step makes from host code NEXT TO a list (after the recording was closed, or while it was paused) and that
    the replay path forgets to repeat is silently missing from every later step -- round 4's batched bias gradients
    (segnb_bias_grad_multi) disappeared that way and every convolution bias without a BatchNorm stopped training.
    With the guard on, the library counts every top-level entry point it executes, by name, whether it was called directly or from
    a replayed list (segnb_tune "call_census" / segnb_debug_census); the census of the step that RECORDED a (model, phase, key)
    is kept and every REPLAYED step of that key must reproduce it exactly, or the step raises with the difference."""
    enabled = os.environ.get('SEGNB_REPLAY_GUARD', '0') != '0'
    _on = False
    IGNORE = ('segnb_tune',)

    @classmethod
    def enable(cls, on=True):
        cls.enabled = bool(on)
        if not on and cls._on:
            nv.call('segnb_tune', b'call_census', 0)
            cls._on = False

    def __init__(self, what):
        self.what = what
        self.ref = {}
        self.checked = 0

    def begin(self):
        if not self.enabled or nv.has_test_backend():
            return False
        if not ReplayGuard._on:
            nv.call('segnb_tune', b'call_census', 1)       # (process-wide switch: never inside a recording -- segnb_tune is refused there)
            ReplayGuard._on = True
        nv.census_read()
        return True

    def end(self, active, key, mode):
        """mode: 'record' (this step recorded the lists of `key`), 'replay' (it ran from them) or anything else (eager: ignored)"""
        if not active:
            return
        c = {k: v for k, v in nv.census_read().items() if k not in self.IGNORE}
        if mode == 'record':
            self.ref[key] = c
        elif mode == 'replay' and key in self.ref:
            ref = self.ref[key]
            if c != ref:
                diff = ['%s: recorded step %d, replayed step %d' % (k, ref.get(k, 0), c.get(k, 0))
                        for k in sorted(set(ref) | set(c)) if ref.get(k, 0) != c.get(k, 0)]
                raise RuntimeError('%s: a replayed step does not execute the launches of the step that recorded it -- %s'
                                   % (self.what, '; '.join(diff)))
            self.checked += 1


class Runtime(object):
    """Per-model execution context: device, compute dtype, stream."""

    def __init__(self, device, dtype='bf16'):
        self.device = torch.device(device)
        if dtype in ('bf16', torch.bfloat16):
            self.code, self.tdtype = nv.BF16, torch.bfloat16
        elif dtype in ('f32', 'fp32', torch.float32):
            self.code, self.tdtype = nv.F32, torch.float32
        else:
            raise ValueError('dtype must be bf16 or f32, got %r' % (dtype,))

    @property
    def stream(self):
        if self.device.type == 'cuda':
            return torch.cuda.current_stream(self.device).cuda_stream
        return 0

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.tdtype, device=self.device)

    # ---- second stream for the weight gradients ------------------------------------------------------------
    # wgrad(l) and dgrad(l) both only READ dy(l), and nothing consumes dW before the end of backward: the weight
    # gradients run on a side stream, forked after each layer's BatchNorm-apply and joined once before the batched
    # unpack.  Neither kernel family fills a CU alone (one wgrad block = 4 waves / 64 KB LDS, two dgrad blocks =
    # 8 waves / 148 KB), so the hardware co-schedules them.  Captured into the step's HIP graph as a parallel branch.
    overlap_wgrad = os.environ.get('SEGNB_OVERLAP_WGRAD', '1') != '0'
    if not overlap_wgrad:
        # one stream: nothing runs beside the weight gradients, so they take every CU (wgrad_s1.hip: s1_slabs reads
        # this once, at the first plan)
        os.environ.setdefault('SEGNB_WG_CU_FRACTION', '1')

    def side_stream(self):
        if self.device.type != 'cuda' or not self.overlap_wgrad:
            return None
        s = getattr(self, '_side', None)
        if s is None:
            s = self._side = torch.cuda.Stream(device=self.device)
        return s

    def arm_fork(self):
        """Call right before the LAST launch a following fork_side() has to wait for: where that launch can carry an event (the
        BatchNorm-backward apply passes) the fork costs the main queue no marker packet (segnb_stream_fork_arm / _commit)."""
        if self.side_stream() is not None:
            nv.call('segnb_stream_fork_arm', self.stream)
            self._armed = True

    def fork_side(self):
        """-> side stream (made to wait for everything issued so far on the current stream) or None"""
        s = self.side_stream()
        if s is not None:
            if getattr(self, '_armed', False):
                nv.call('segnb_stream_fork_commit', self.stream, s.cuda_stream)
                self._armed = False
            else:
                nv.call('segnb_stream_fork', self.stream, s.cuda_stream)   # (an ABI call: recordable in a launch plan)
            self._side_busy = True
        return s

    def join_side(self):
        self._armed = False                  # (an armed fork that never committed -- an exception in between -- dies here)
        if getattr(self, '_side_busy', False):
            nv.call('segnb_stream_join', self.stream, self._side.cuda_stream)
            self._side_busy = False

    def int32(self, values):
        return torch.tensor(list(values), dtype=torch.int32, device=self.device)

    # clears / copies of (strided) activation views as ABI launches, so that a recorded launch list contains them
    def clear_view(self, v):
        nv.call('segnb_add', self.code, None, 0, None, 0, v.ptr, v.ld, v.N, v.H, v.W, v.Cp, self.stream)

    def copy_view(self, src, dst):
        nv.call('segnb_add', self.code, None, 0, src.ptr, src.ld, dst.ptr, dst.ld, dst.N, dst.H, dst.W, dst.Cp, self.stream)


class InputNorm(object):
    """NormalizeImage of the reference's input pipeline (lib/augmentations.py:452-460): x * scale - mean) / std per
    channel, applied on the device to a uint8 HWC batch.  Defaults = the reference's."""

    def __init__(self, scale=1. / 255., mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.scale, self.mean, self.std = float(scale), tuple(float(m) for m in mean), tuple(float(v) for v in std)
        if len(self.mean) != len(self.std) or not all(v != 0.0 for v in self.std):
            raise ValueError('mean / std must have one non-zero std per channel')

    def arrays(self, C):
        if len(self.mean) != C:
            raise ValueError('input has %d channels, normalisation has %d' % (C, len(self.mean)))
        return nv.float_array(self.mean), nv.float_array(self.std)


def pack_input(rt, x, xv, norm=None):
    """The network input -> the NHWC `rt.tdtype` view xv (channels zero-padded): float32 NCHW (what torch_train.py:177
    hands the model) or uint8 NHWC (what the dataset holds, normalised on the device with `norm`)."""
    if x.dtype == torch.uint8:
        N, H, W, C = x.shape
        mean, std = (norm or InputNorm()).arrays(C)
        nv.call('segnb_pack_input_u8', nv.ptr(x), N, H, W, C, (norm or InputNorm()).scale, mean, std, xv.ptr, rt.code,
                xv.Cp, xv.ld, rt.stream)
    else:
        N, C, H, W = x.shape
        nv.call('segnb_pack_input_nchw', nv.ptr(x), N, C, H, W, xv.ptr, rt.code, xv.Cp, xv.ld, rt.stream)


class View(object):
    """[N, H, W, Cp] NHWC activation = channel slice [off, off+Cp) of a buffer with pixel stride ld."""
    __slots__ = ('t', 'off', 'ld', 'N', 'H', 'W', 'Cp')

    def __init__(self, t, N, H, W, Cp, ld=None, off=0):
        self.t, self.N, self.H, self.W, self.Cp = t, N, H, W, Cp
        self.ld = Cp if ld is None else ld
        self.off = off

    @staticmethod
    def alloc(rt, N, H, W, Cp):
        return View(rt.zeros((N, H, W, Cp)), N, H, W, Cp)

    def slice(self, off, Cp):
        return View(self.t, self.N, self.H, self.W, Cp, self.ld, self.off + off)

    @property
    def ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()

    def dense(self):
        """torch view [N,H,W,Cp] (debug / tests)."""
        return self.t.view(self.N, self.H, self.W, self.ld)[..., self.off:self.off + self.Cp]


def vptr(v):
    return None if v is None else v.ptr


def vld(v):
    return 0 if v is None else v.ld


# Set by a model's backward driver for the duration of one backward: True when FlatParams.begin_backward() cleared the flat
# gradient buffer (no .grad aliased it), i.e. every directly delivered weight gradient may be STORED; False (the default, and
# whenever gradients accumulate in place across backward calls) = added.  Part of the recorded backward lists' keys.
DW_OVERWRITE = False


class ConvOp(object):
    """One nn.Conv2d / nn.ConvTranspose2d parameter set on the gather-conv kernels.

    in_segments: [(real_channels, padded_channels), ...] -- the channel layout of the NHWC input view
    (a torch.cat of padded slices); real channels of consecutive segments are consecutive in the
    reference's weight tensor.
    """

    # (Leaving the weight-gradient slabs unreduced for the batched unpack to sum -- segnb_conv_wgrad_partial, round 2 -- measured
    # 5.63 ms/step against 5.39: a layer with one channel tile has up to 128 slabs that one unpack thread walks serially on the
    # critical path; removed in round 5.)

    algo_scale = 1.0        # algorithmic / executed FLOPs of a launch (UpConvOp: 9 / 4)
    _in_place, _ci_offset = False, 0      # (subclasses that build their own channel maps keep the workspace + unpack path)
    pack_fwd = True         # False: the forward matrix is never used (the owner runs the forward through another op)

    def __init__(self, rt, weight, bias, in_segments, stride=1, pad=1, transposed=False, need_dgrad=True,
                 out_hw=None, ci_offset=0):
        """ci_offset: the op covers input channels [ci_offset, ci_offset + sum(real)) of the parameter tensor only (one
        segment of a concatenated input handled on its own)."""
        self.rt = rt
        self.out_hw_override = out_hw      # transposed conv only: crop the output at the bottom/right
                                           # (center_crop of tiramisu.py:86-90 always has offset 0)
        self.weight, self.bias = weight, bias
        self.stride, self.pad, self.transposed = stride, pad, transposed
        self.need_dgrad = need_dgrad
        if transposed:
            self.Ci, self.Co, self.KH, self.KW = weight.shape
            self.s_out, self.s_in = self.KH * self.KW, self.Co * self.KH * self.KW
        else:
            self.Co, self.Ci, self.KH, self.KW = weight.shape
            self.s_out, self.s_in = self.Ci * self.KH * self.KW, self.KH * self.KW
        if ci_offset or sum(r for r, _ in in_segments) != self.Ci:
            assert not transposed and ci_offset + sum(r for r, _ in in_segments) <= self.Ci
            self.Ci = sum(r for r, _ in in_segments)       # (strides above stay those of the whole parameter)
        assert sum(r for r, _ in in_segments) == self.Ci, (in_segments, self.Ci)
        self.Cop = cp.pad8(self.Co)
        imap = []
        base = ci_offset
        for real, padded in in_segments:
            assert padded % 8 == 0 and padded >= real
            imap += [base + i for i in range(real)] + [-1] * (padded - real)
            base += real
        self.Cip = len(imap)
        self.in_map = rt.int32(imap)
        self.out_map = rt.int32(list(range(self.Co)) + [-1] * (self.Cop - self.Co))
        self._plans = {}
        # real input channel of padded channel j is ci_offset + j (padding only behind the real channels): the weight gradient
        # can be delivered straight into the parameter's gradient (segnb_wgrad_target), no workspace unpack
        self._in_place = (not transposed and
                          all(m == ci_offset + j for j, m in enumerate(imap) if m >= 0) and
                          all(m < 0 for m in imap[self.Ci:]))
        self._ci_offset = ci_offset

    # ---- per-input-size plan: launches, packed buffers -----------------------------------------
    def plan(self, Hi, Wi):
        key = (Hi, Wi)
        p = self._plans.get(key)
        if p is not None:
            return p
        rt = self.rt
        p = {}
        if self.transposed:
            (Ho, Wo), fwd, full = cp.convt_fwd(Hi, Wi, self.KH, self.KW, self.stride, self.pad)
            if self.out_hw_override is not None:
                Hc, Wc = self.out_hw_override(Hi, Wi) if callable(self.out_hw_override) else self.out_hw_override
                assert Hc <= Ho and Wc <= Wo, 'crop must not exceed the transposed-conv output'
                fwd, full = cp._scatter_phases(Hc, Wc, Hi, Wi, self.KH, self.KW, self.stride, self.pad)
                Ho, Wo = Hc, Wc
            dg, dg_full = cp.convt_dgrad(Hi, Wi, self.KH, self.KW, self.stride, self.pad), True
        else:
            (Ho, Wo), fwd = cp.conv_fwd(Hi, Wi, self.KH, self.KW, self.stride, self.pad)
            full = True
            dg, dg_full = cp.conv_dgrad(Hi, Wi, self.KH, self.KW, self.stride, self.pad)
        p['out_hw'] = (Ho, Wo)
        p['fwd'], p['fwd_full'] = fwd, full
        p['dg'], p['dg_full'] = dg, dg_full
        p['wp_fwd'] = [rt.zeros((self.Cop, len(l.taps) * self.Cip)) for l in fwd]
        if self.transposed and (self.KH, self.KW, self.stride, self.pad) == (4, 4, 2, 1) and len(fwd) == 4 and full and \
                self.out_hw_override is None:
            # ConvTranspose2d(4, 2, 1): the four phase matrices in ONE buffer [4][Cop][4 * Cip] -- segnb_upconv_fprop runs the
            # phases as one launch where it serves the shape
            p['wp_fwd_all'] = rt.zeros((4, self.Cop, 4 * self.Cip))
            p['wp_fwd'] = [p['wp_fwd_all'][i] for i in range(4)]
        p['tapoff_fwd'] = [nv.int_array([a * self.KW + b for (_, _, a, b) in l.taps]) for l in fwd]
        if self.need_dgrad:
            p['wp_dg'] = [rt.zeros((self.Cip, len(l.taps) * self.Cop)) for l in dg]
            p['tapoff_dg'] = [nv.int_array([a * self.KW + b for (_, _, a, b) in l.taps]) for l in dg]
        # weight-gradient workspace (fp32, packed like the forward matrix): nslab partial slabs per launch, as
        # many as segnb_conv_wgrad_slabs asks for (a property of the channel counts, taps and width -- not of the
        # batch).  The result is always slab 0 (zeroed here once, re-zeroed by segnb_unpack_wgrad when consumed);
        # slabs 1.. are scratch for the partial sums of the pixel ranges.
        if self.transposed:
            geoms = [self._make_geom(l, 1, Ho, Wo, self.Cop, self.Cop, Hi, Wi, self.Cip, self.Cip) for l in dg]
            p['nslab'] = self._under_wg_share(lambda: [nv.query('segnb_conv_wgrad_slabs', g, rt.code) for g in geoms])
            p['dwp'] = [rt.zeros((n, self.Cip, len(l.taps) * self.Cop), torch.float32)
                        for n, l in zip(p['nslab'], dg)]
        else:
            geoms = [self._make_geom(l, 1, Hi, Wi, self.Cip, self.Cip, Ho, Wo, self.Cop, self.Cop) for l in fwd]
            p['nslab'] = self._under_wg_share(lambda: [nv.query('segnb_conv_wgrad_slabs', g, rt.code) for g in geoms])
            p['dwp'] = [rt.zeros((n, self.Cop, len(l.taps) * self.Cip), torch.float32)
                        for n, l in zip(p['nslab'], fwd)]
        p['geoms'] = {}
        self._plans[key] = p
        return p

    def out_hw(self, Hi, Wi):
        return self.plan(Hi, Wi)['out_hw']

    def _geom(self, p, tag, li, launch, N, Hi, Wi, Ci, ld_in, Ho, Wo, Co, ld_out):
        key = (tag, li, N, ld_in, ld_out)
        g = p['geoms'].get(key)
        if g is None:
            g = p['geoms'][key] = self._make_geom(launch, N, Hi, Wi, Ci, ld_in, Ho, Wo, Co, ld_out)
        return g

    @staticmethod
    def _make_geom(launch, N, Hi, Wi, Ci, ld_in, Ho, Wo, Co, ld_out):
        g = nv.ConvGeom()
        g.N, g.Hi, g.Wi, g.Ci = N, Hi, Wi, Ci
        g.Ho, g.Wo, g.Co = Ho, Wo, Co
        g.ld_in, g.ld_out = ld_in, ld_out
        g.QH, g.QW = launch.QH, launch.QW
        g.in_step, g.out_step = launch.in_step, launch.out_step
        g.oh0, g.ow0 = launch.oh0, launch.ow0
        g.ntaps = len(launch.taps)
        if g.ntaps > nv.MAX_TAPS:
            raise ValueError('kernel has more than %d taps' % nv.MAX_TAPS)
        for i, (dh, dw, _, _) in enumerate(launch.taps):
            g.dh[i], g.dw[i] = dh, dw
        return g

    # ---- weight packing (every time the parameters changed) ---------------------------------------
    def pack(self, Hi, Wi):
        p, rt = self.plan(Hi, Wi), self.rt
        w = self.weight.detach()
        for li, l in enumerate(p['fwd']):
            nv.call('segnb_pack_weight', nv.ptr(w), nv.ptr(p['wp_fwd'][li]), rt.code, self.Cop, self.Cip,
                    len(l.taps), self.s_out, self.s_in, p['tapoff_fwd'][li], nv.ptr(self.out_map),
                    nv.ptr(self.in_map), rt.stream)
        if self.need_dgrad:
            for li, l in enumerate(p['dg']):
                nv.call('segnb_pack_weight', nv.ptr(w), nv.ptr(p['wp_dg'][li]), rt.code, self.Cip, self.Cop,
                        len(l.taps), self.s_in, self.s_out, p['tapoff_dg'][li], nv.ptr(self.in_map),
                        nv.ptr(self.out_map), rt.stream)

    def pack_jobs(self, Hi, Wi):
        """Job records (PackTable) equivalent to pack(Hi, Wi)."""
        p, rt = self.plan(Hi, Wi), self.rt
        w = self.weight.detach()
        jobs = []
        for li, l in enumerate(p['fwd'] if self.pack_fwd else ()):
            jobs.append(dict(w=w, packed=p['wp_fwd'][li], mmap=self.out_map, cmap=self.in_map, s_m=self.s_out,
                             s_c=self.s_in, Mp=self.Cop, Cp=self.Cip, ntaps=len(l.taps), dtype=rt.code, form='f',
                             tap_off=[a * self.KW + b for (_, _, a, b) in l.taps]))
        if self.need_dgrad:
            for li, l in enumerate(p['dg']):
                jobs.append(dict(w=w, packed=p['wp_dg'][li], mmap=self.in_map, cmap=self.out_map, s_m=self.s_in,
                                 s_c=self.s_out, Mp=self.Cip, Cp=self.Cop, ntaps=len(l.taps), dtype=rt.code, form='d',
                                 tap_off=[a * self.KW + b for (_, _, a, b) in l.taps]))
        return jobs

    def unpack_jobs(self, Hi, Wi, grad_w):
        """Job records equivalent to the segnb_unpack_wgrad calls of wgrad(); use with wgrad(..., unpack=False)."""
        p = self.plan(Hi, Wi)
        jobs = []
        if self.direct_ok():
            return jobs          # (delivered by the weight-gradient launches themselves: segnb_wgrad_target)
        if self.transposed:
            for li, l in enumerate(p['dg']):
                jobs.append(dict(w=grad_w, packed=p['dwp'][li], mmap=self.in_map, cmap=self.out_map, s_m=self.s_in,
                                 s_c=self.s_out, Mp=self.Cip, Cp=self.Cop, ntaps=len(l.taps), dtype=nv.F32,
                                 nslab=1, tap_off=[a * self.KW + b for (_, _, a, b) in l.taps]))
        else:
            for li, l in enumerate(p['fwd']):
                jobs.append(dict(w=grad_w, packed=p['dwp'][li], mmap=self.out_map, cmap=self.in_map, s_m=self.s_out,
                                 s_c=self.s_in, Mp=self.Cop, Cp=self.Cip, ntaps=len(l.taps), dtype=nv.F32,
                                 nslab=1, tap_off=[a * self.KW + b for (_, _, a, b) in l.taps]))
        return jobs

    # ---- kernels ------------------------------------------------------------------------------------
    # activation (and eval-mode BatchNorm) in the convolution's epilogue instead of a pass of its own (A/B: class attribute fuse_act)
    fuse_act = True

    def fprop(self, xv, yv, stats=None, epilogue=None):
        """epilogue: (coef or None, act, slope) -- segnb_conv_fprop_act: yv receives act(conv + bias) (coef None) or
        act of the eval-mode BatchNorm of it (coef = [4][Cop] of segnb_bn_finalize); no statistics then."""
        p, rt = self.plan(xv.H, xv.W), self.rt
        assert xv.Cp == self.Cip and yv.Cp == self.Cop, (xv.Cp, self.Cip, yv.Cp, self.Cop)
        assert (yv.H, yv.W) == p['out_hw']
        if not p['fwd_full']:
            rt.clear_view(yv)
        b = self.bias.detach() if self.bias is not None else None
        if epilogue is not None:
            # (the phase launches of a transposed convolution write disjoint output parities: each applies the epilogue to its own)
            assert stats is None and p['fwd_full'], 'full coverage, no statistics'
            ep = nv.ActEpilogue(nv.ptr(epilogue[0]), epilogue[1], epilogue[2])
            if 'wp_fwd_all' in p:
                # ConvTranspose2d(4, 2, 1): the four phases as one launch, the activation in its accumulator staging
                assert epilogue[0] is None and self.upconv_act_ok(xv.N, xv.H, xv.W, yv.ld)
                _timed('conv_fprop', 2.0 * xv.N * xv.H * xv.W * 16 * self.Ci * self.Co,
                       lambda: nv.call('segnb_upconv_fprop_act', rt.code, xv.N, xv.H, xv.W, self.Cip, xv.ld, xv.ptr,
                                       nv.ptr(p['wp_fwd_all']), self.Cop, self.Cop, nv.ptr(b), self.Co if b is not None else 0,
                                       yv.ptr, yv.ld, ep, rt.stream))
                return
            for li, l in enumerate(p['fwd']):
                g = self._geom(p, 'f', li, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)
                _timed('conv_fprop', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                       lambda: nv.call('segnb_conv_fprop_act', g, rt.code, xv.ptr, nv.ptr(p['wp_fwd'][li]), nv.ptr(b),
                                       self.Co if b is not None else 0, yv.ptr, ep, rt.stream))
            return
        if type(self) is ConvOp and 'wp_fwd_all' in p and \
                nv.query('segnb_upconv_fprop_ok', xv.N, xv.H, xv.W, self.Cip, self.Cop, yv.ld, rt.code):
            ex = 2.0 * xv.N * xv.H * xv.W * 16 * self.Ci * self.Co
            _timed('conv_fprop', ex,
                   lambda: nv.call('segnb_upconv_fprop', rt.code, xv.N, xv.H, xv.W, self.Cip, xv.ld, xv.ptr,
                                   nv.ptr(p['wp_fwd_all']), self.Cop, self.Cop, nv.ptr(b), self.Co if b is not None else 0,
                                   yv.ptr, yv.ld, nv.ptr(stats), rt.stream))
            return
        for li, l in enumerate(p['fwd']):
            g = self._geom(p, 'f', li, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)
            ex = 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co
            _timed('conv_fprop', ex * self.algo_scale,
                   lambda: nv.call('segnb_conv_fprop', g, rt.code, xv.ptr, nv.ptr(p['wp_fwd'][li]), nv.ptr(b),
                                   self.Co if b is not None else 0, yv.ptr, nv.ptr(stats), rt.stream), ex)

    def upconv_act_ok(self, N, H, W, ld_out):
        p = self.plan(H, W)
        return (type(self) is ConvOp and 'wp_fwd_all' in p
                and bool(nv.query('segnb_upconv_fprop_ok', N, H, W, self.Cip, self.Cop, ld_out, self.rt.code)))

    def act_epilogue_ok(self, H, W, N=None, ld_out=None, coef=None):
        """segnb_conv_fprop_act: one launch that covers the whole output, or the parity phases of a transposed convolution (disjoint
        outputs, together everything: linknet.py:58 finaldeconv1).  A ConvTranspose2d(4, 2, 1) whose four phases run as ONE launch of
        the direct-to-LDS kernel takes a plain activation there (segnb_upconv_fprop_act; needs N and the output stride) and
        otherwise keeps the separate activation pass (it costs less than four launches of the general kernel)."""
        p = self.plan(H, W)
        if not (self.fuse_act and p['fwd_full']):
            return False
        if len(p['fwd']) == 1:
            return True
        if not self.transposed:
            return False
        if 'wp_fwd_all' in p:
            return N is not None and coef is None and self.upconv_act_ok(N, H, W, ld_out)
        return True

    # fuse_drop = False (class attribute / SEGNB_FUSE_DROP=0): conv -> Dropout2d -> slice statistics as convolution + pass (A/B)
    fuse_drop = os.environ.get('SEGNB_FUSE_DROP', '1') != '0'

    def drop_epilogue_ok(self, N, H, W, ld_out):
        """True when segnb_conv_fprop_drop serves this convolution: the Dropout2d multipliers (and the statistics of the result) in
        the launch's store pass."""
        p = self.plan(H, W)
        if not self.fuse_drop or self.transposed or len(p['fwd']) != 1 or not p['fwd_full'] or self.rt.code != nv.BF16:
            return False
        Ho, Wo = p['out_hw']
        g = self._geom(p, 'f', 0, p['fwd'][0], N, H, W, self.Cip, self.Cip, Ho, Wo, self.Cop, ld_out)
        return bool(nv.query('segnb_conv_fprop_drop_ok', g, self.rt.code))

    def fprop_drop(self, xv, yv, dropmul, out_stats=None):
        """yv = Dropout2d multipliers x (conv + bias); out_stats: (fp64 table, element offset, row stride) -- the statistics of yv are
        accumulated into that channel range -- or None.  Check drop_epilogue_ok first."""
        p, rt = self.plan(xv.H, xv.W), self.rt
        assert xv.Cp == self.Cip and yv.Cp == self.Cop and (yv.H, yv.W) == p['out_hw'] and dropmul.shape[1] >= self.Cop
        l = p['fwd'][0]
        g = self._geom(p, 'f', 0, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)
        b = self.bias.detach() if self.bias is not None else None
        st = nv.ptr(out_stats[0], out_stats[1]) if out_stats is not None else None
        _timed('conv_fprop', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
               lambda: nv.call('segnb_conv_fprop_drop', g, rt.code, xv.ptr, nv.ptr(p['wp_fwd'][0]), nv.ptr(b),
                               self.Co if b is not None else 0, yv.ptr, nv.ptr(dropmul), int(dropmul.shape[1]), st,
                               out_stats[2] if out_stats is not None else 0, rt.stream))

    def u8_direct_ok(self, N, H, W, ld_out):
        """True when segnb_conv_fprop_u8 serves this convolution as the network's first layer."""
        p = self.plan(H, W)
        if len(p['fwd']) != 1 or self.Cip != 8 or p['out_hw'] != (H, W):
            return False
        g = self._geom(p, 'f', 0, p['fwd'][0], N, H, W, self.Cip, self.Cip, H, W, self.Cop, ld_out)
        return bool(nv.query('segnb_conv_fprop_u8_ok', g, self.rt.code))

    def fprop_u8(self, img, norm, yv, stats=None, packed=None):
        """First layer straight from the uint8 HWC batch `img` [N,H,W,C]; packed: optional View that receives the
        normalised bf16 pixels (the x operand of this layer's weight gradient)."""
        N, H, W, C = img.shape
        p, rt = self.plan(H, W), self.rt
        l = p['fwd'][0]
        g = self._geom(p, 'f', 0, l, N, H, W, self.Cip, self.Cip, yv.H, yv.W, self.Cop, yv.ld)
        mean, std = norm.arrays(C)
        b = self.bias.detach() if self.bias is not None else None
        _timed('conv_fprop', 2.0 * N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
               lambda: nv.call('segnb_conv_fprop_u8', g, nv.ptr(img), C, norm.scale, mean, std, nv.ptr(p['wp_fwd'][0]),
                               nv.ptr(b), self.Co if b is not None else 0, yv.ptr, nv.ptr(stats), vptr(packed),
                               vld(packed), rt.stream))

    def dgrad_bnreduce_ok(self, dyv, dxv):
        """True when this data gradient can also do the BatchNorm-backward reduction of the layer that produced its
        input (segnb_conv_fprop_bnreduce): one launch geometry, served by a fused kernel."""
        p = self.plan(dxv.H, dxv.W)
        if not self.need_dgrad or len(p['dg']) != 1 or not p['dg_full']:
            return False
        g = self._geom(p, 'd', 0, p['dg'][0], dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, dxv.H, dxv.W, self.Cip, dxv.ld)
        return bool(nv.query('segnb_conv_fprop_bnreduce_ok', g, self.rt.code))

    def dgrad_bnapply_ok(self, dyv, N, H, W, ld):
        """True when this data gradient can run as the two launches that never store it (segnb_conv_fprop_bnsums / _bnapply: a dense
        layer's 16 -> prefix gradient); N, H, W, ld: the input tensor's geometry (the gradient itself gets no buffer)."""
        p = self.plan(H, W)
        if not self.need_dgrad or len(p['dg']) != 1 or not p['dg_full']:
            return False
        g = self._geom(p, 'd', 0, p['dg'][0], dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, H, W, self.Cip, ld)
        return bool(nv.query('segnb_conv_fprop_bnapply_ok', g, self.rt.code))

    def dgrad_bnsums(self, dyv, H, W, bn_reduce):
        """first launch: the BatchNorm-backward sums of the layer whose activation gradient this data gradient is; nothing stored"""
        p, rt = self.plan(H, W), self.rt
        l = p['dg'][0]
        yv, coef, sums, act, slope = bn_reduce
        g = self._geom(p, 'd', 0, l, dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, H, W, self.Cip, yv.ld)
        ep = nv.BnReduceEpilogue(yv.ptr, yv.ld, nv.ptr(coef), nv.ptr(sums), act, slope)
        _timed('conv_fprop', 2.0 * dyv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
               lambda: nv.call('segnb_conv_fprop_bnsums', g, rt.code, dyv.ptr, nv.ptr(p['wp_dg'][0]), ep, rt.stream))

    def dgrad_bnapply(self, dyv, H, W, ep):
        """second launch (ep: nv.BnApplyEpilogue): the gradient recomputed, BatchNorm backward applied, written / added to ep.dx"""
        p, rt = self.plan(H, W), self.rt
        l = p['dg'][0]
        g = self._geom(p, 'd', 0, l, dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, H, W, self.Cip, ep.ld_dx)
        _timed('conv_fprop', 2.0 * dyv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
               lambda: nv.call('segnb_conv_fprop_bnapply', g, rt.code, dyv.ptr, nv.ptr(p['wp_dg'][0]), ep, rt.stream))

    def dgrad_actmask_ok(self, dyv, dxv):
        """True when this data gradient can apply the activation mask of the conv + activation (no BatchNorm) that produced its
        input and store dz (segnb_conv_fprop_bnreduce with coef None)."""
        p = self.plan(dxv.H, dxv.W)
        if not self.need_dgrad or len(p['dg']) != 1 or not p['dg_full']:
            return False
        g = self._geom(p, 'd', 0, p['dg'][0], dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, dxv.H, dxv.W, self.Cip, dxv.ld)
        return bool(nv.query('segnb_conv_fprop_actmask_ok', g, self.rt.code))

    def dgrad(self, dyv, dxv, bn_reduce=None):
        """bn_reduce: (y View, coef, sums, act, slope) of the layer whose activation gradient dxv is -- its reduction
        pass is then done by this launch's epilogue (check dgrad_bnreduce_ok first; coef None: the activation mask of a layer
        without BatchNorm, dxv receives dz -- check dgrad_actmask_ok)."""
        p, rt = self.plan(dxv.H, dxv.W), self.rt
        assert self.need_dgrad and dyv.Cp == self.Cop and dxv.Cp == self.Cip
        if not p['dg_full']:
            rt.clear_view(dxv)
        for li, l in enumerate(p['dg']):
            g = self._geom(p, 'd', li, l, dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, dxv.H, dxv.W, self.Cip, dxv.ld)
            if bn_reduce is not None:
                yv, coef, sums, act, slope = bn_reduce
                ep = nv.BnReduceEpilogue(yv.ptr, yv.ld, nv.ptr(coef), nv.ptr(sums), act, slope)
                _timed('conv_fprop', 2.0 * dyv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                       lambda: nv.call('segnb_conv_fprop_bnreduce', g, rt.code, dyv.ptr, nv.ptr(p['wp_dg'][li]), dxv.ptr,
                                       ep, rt.stream))
                continue
            ex = 2.0 * dyv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co
            _timed('conv_fprop', ex * self.algo_scale,
                   lambda: nv.call('segnb_conv_fprop', g, rt.code, dyv.ptr, nv.ptr(p['wp_dg'][li]), None, 0,
                                   dxv.ptr, None, rt.stream), ex)

    # ---- operands recomputed on load (segnb_operand_tf: consumer-side BatchNorm, fprop_roll.hip / wgrad_roll.hip) -------------
    @staticmethod
    def tf_act(coef, Cp, act, slope, drop=None):
        """operand = round(drop * act(BatchNorm(src))): the activation pass of the producing layer, applied by the consumer"""
        return nv.OperandTf(nv.TF_ACT, None, 0, nv.ptr(coef), None, nv.ptr(drop), Cp, act, slope)

    @staticmethod
    def tf_bnbwd(yv, coef, bcoef, act, slope):
        """operand = dy = BatchNorm-backward apply of (src, yv); act = ACT_NONE when src is dz already"""
        return nv.OperandTf(nv.TF_BNBWD, yv.ptr, yv.ld, nv.ptr(coef), nv.ptr(bcoef), None, yv.Cp, act, slope)

    def fprop_tf_ok(self, xv, yv):
        p = self.plan(xv.H, xv.W)
        if self.transposed or len(p['fwd']) != 1 or not p['fwd_full']:
            return False
        g = self._geom(p, 'f', 0, p['fwd'][0], xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)
        return bool(nv.query('segnb_conv_fprop_tf_ok', g, self.rt.code, nv.TF_ACT))

    def fprop_tf(self, xv, tf, yv, stats=None):
        """forward whose input operand is tf(xv) (xv: the producing layer's pre-BatchNorm output)"""
        p, rt = self.plan(xv.H, xv.W), self.rt
        l = p['fwd'][0]
        g = self._geom(p, 'f', 0, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)
        b = self.bias.detach() if self.bias is not None else None
        _timed('conv_fprop', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
               lambda: nv.call('segnb_conv_fprop_tf', g, rt.code, xv.ptr, tf, nv.ptr(p['wp_fwd'][0]), nv.ptr(b),
                               self.Co if b is not None else 0, yv.ptr, nv.ptr(stats), None, rt.stream))

    def dgrad_tf_ok(self, dyv, dxv):
        p = self.plan(dxv.H, dxv.W)
        if not self.need_dgrad or len(p['dg']) != 1 or not p['dg_full']:
            return False
        g = self._geom(p, 'd', 0, p['dg'][0], dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, dxv.H, dxv.W, self.Cip, dxv.ld)
        return bool(nv.query('segnb_conv_fprop_tf_ok', g, self.rt.code, nv.TF_BNBWD))

    def dgrad_tf(self, gv, tf, dxv, bn_reduce=None):
        """data gradient whose dy operand is tf(gv) (gv: the gradient of this layer's activation, or dz); bn_reduce as dgrad()"""
        p, rt = self.plan(dxv.H, dxv.W), self.rt
        l = p['dg'][0]
        g = self._geom(p, 'd', 0, l, gv.N, gv.H, gv.W, self.Cop, gv.ld, dxv.H, dxv.W, self.Cip, dxv.ld)
        ep = None
        if bn_reduce is not None:
            yv, coef, sums, act, slope = bn_reduce
            ep = nv.BnReduceEpilogue(yv.ptr, yv.ld, nv.ptr(coef), nv.ptr(sums), act, slope)
        _timed('conv_fprop', 2.0 * gv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
               lambda: nv.call('segnb_conv_fprop_tf', g, rt.code, gv.ptr, tf, nv.ptr(p['wp_dg'][0]), None, 0, dxv.ptr, None, ep,
                               rt.stream))

    def wgrad_tf_ok(self, xv, dyv):
        p = self.plan(xv.H, xv.W)
        if self.transposed or len(p['fwd']) != 1:
            return False
        g = self._geom(p, 'f', 0, p['fwd'][0], xv.N, xv.H, xv.W, self.Cip, xv.ld, dyv.H, dyv.W, self.Cop, dyv.ld)
        return bool(nv.query('segnb_conv_wgrad_tf_ok', g, self.rt.code))

    def wgrad_tf(self, xv, tfx, dv, tfd, grad_w=None):
        """weight gradient with x = tfx(xv) and dy = tfd(dv) (either transform may be None); the result is left in the
        packed workspace for the batched unpack, like wgrad(..., unpack=False), or delivered into grad_w (direct_ok)"""
        p, rt = self.plan(xv.H, xv.W), self.rt
        l = p['fwd'][0]
        g = self._geom(p, 'f', 0, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, dv.H, dv.W, self.Cop, dv.ld)

        def launch():
            if self.direct_ok() and grad_w is not None:
                self._arm_target(p, 0, grad_w)
            nv.call('segnb_conv_wgrad_tf', g, rt.code, xv.ptr, tfx, dv.ptr, tfd, nv.ptr(p['dwp'][0]), p['nslab'][0], rt.stream)
        _timed('conv_wgrad', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co, lambda: self._under_wg_share(launch))

    # share (%) of the CUs this convolution's weight gradient splits its pixels for; None = the library default.  Set before
    # the first plan (the workspace is sized under it); every launch is bracketed by segnb_wg_cu_share (recordable)
    wg_cu_pct = None

    def _under_wg_share(self, fn):
        if self.wg_cu_pct is None:
            return fn()
        nv.call('segnb_wg_cu_share', int(self.wg_cu_pct))
        try:
            return fn()
        finally:
            nv.call('segnb_wg_cu_share', 0)

    def wgrad_bnapply_ok(self, xv, yv):
        """True when this convolution's weight gradient can recompute its dy operand -- the BatchNorm-backward apply of the
        layer -- from (g, y) itself (segnb_conv_wgrad_bnapply): the apply pass then disappears for a layer without a data
        gradient.  By default only the first layer's rolling kernel (8 padded input channels) takes it: 52.6 us against
        52.5 us (apply pass) + 58.8 us (weight gradient on the stored dz) stand-alone, -0.85 % of the timed step;
        SEGNB_WGRAD_BNAPPLY=1 adds the thin tile kernel's variant (measured neutral), =0 disables both."""
        p = self.plan(xv.H, xv.W)
        if self.transposed or len(p['fwd']) != 1:
            return False
        g = self._geom(p, 'f', 0, p['fwd'][0], xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)
        return bool(nv.query('segnb_conv_wgrad_bnapply_ok', g, self.rt.code))

    def wgrad_bnapply(self, xv, gv, yv, coef, bcoef, act, slope, grad_w=None):
        """the weight gradient with dy = BatchNorm-backward apply of (gv, yv); result left in the packed workspace, or delivered
        into grad_w (direct_ok)"""
        p, rt = self.plan(xv.H, xv.W), self.rt
        l = p['fwd'][0]
        g = self._geom(p, 'f', 0, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, yv.H, yv.W, self.Cop, yv.ld)

        def launch():
            if self.direct_ok() and grad_w is not None:
                self._arm_target(p, 0, grad_w)
            nv.call('segnb_conv_wgrad_bnapply', g, rt.code, xv.ptr, gv.ptr, gv.ld, yv.ptr, yv.ld, nv.ptr(coef), nv.ptr(bcoef),
                    self.Cop, act, slope, nv.ptr(p['dwp'][0]), p['nslab'][0], rt.stream)
        _timed('conv_wgrad', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co, lambda: self._under_wg_share(launch))

    # direct_dw = False (class attribute / SEGNB_DIRECT_DW=0): weight gradients through the packed workspace + batched unpack (A/B)
    direct_dw = os.environ.get('SEGNB_DIRECT_DW', '1') != '0'

    # (only for parameters above a size -- the thin layers' 9 - 36 k weights through the workspace + batched unpack -- measured the same:
    # ZF_UNET 4.729 / 4.736 / 4.746 ms at thresholds 0 / 30 k / 300 k, FCDenseNet103 and LinkNet34 +-0.2 %, profiles/r06_ab.txt)
    direct_min_numel = 0

    # dw_store = False (class attribute / SEGNB_DW_STORE=0): directly delivered weight gradients are always read-modify-written (A/B)
    dw_store = os.environ.get('SEGNB_DW_STORE', '1') != '0'

    def direct_ok(self):
        """Do this convolution's weight-gradient launches deliver into the parameter's gradient themselves?"""
        return bool(self.direct_dw and self._in_place and self.rt.code == nv.BF16 and self.weight.numel() >= self.direct_min_numel)

    def _arm_target(self, p, li, grad_w):
        """segnb_wgrad_target_arm for forward launch li: the next weight-gradient call adds its result to grad_w (the fp32
        gradient of the whole parameter, reference layout [Co][Ci_total][KH][KW]) -- or STORES it when the backward that is
        running said the flat gradient buffer holds fresh zeros (DW_OVERWRITE: 0 + x == x, so the read half of the
        read-modify-write -- a dependent round trip per output row of the delivering kernels -- is dropped; a forward
        convolution has exactly one weight-gradient launch)"""
        over = bool(DW_OVERWRITE and self.dw_store and len(p['fwd']) == 1)
        key = ('tgt', li, grad_w.data_ptr(), over)
        t = p.get(key)
        if t is None:
            l = p['fwd'][li]
            t = nv.WgradTarget()
            t.gw = grad_w.data_ptr()
            t.s_out, t.s_in, t.ci_off = self.s_out, self.s_in, self._ci_offset
            t.Ci, t.Co, t.accumulate, t.ntaps = self.Ci, self.Co, 0 if over else 1, len(l.taps)
            for i, (_, _, a, b) in enumerate(l.taps):
                t.kpos[i] = a * self.KW + b
            p[key] = t
        nv.call('segnb_wgrad_target_arm', t)

    def wgrad(self, xv, dyv, grad_w, unpack=True):
        """dW accumulated into grad_w (fp32, parameter layout).  unpack=False leaves the result in the packed
        workspace (slab 0) for a later batched segnb_unpack_wgrad_multi (unpack_jobs) -- unless the launches deliver into
        grad_w themselves (direct_ok: no unpack job exists for them)."""
        p, rt = self.plan(xv.H, xv.W), self.rt
        gw = grad_w
        entry = 'segnb_conv_wgrad'
        if self.direct_ok():
            for li, l in enumerate(p['fwd']):
                g = self._geom(p, 'f', li, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, dyv.H, dyv.W, self.Cop, dyv.ld)

                def launch():
                    self._arm_target(p, li, gw)
                    nv.call(entry, g, rt.code, xv.ptr, dyv.ptr, nv.ptr(p['dwp'][li]), p['nslab'][li], rt.stream)
                _timed('conv_wgrad', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                       lambda: self._under_wg_share(launch))
            return
        if self.transposed:
            # dW[ci][co][k] = sum_hi x[hi][ci] * dy[hi*s - pad + k][co]: "dout" := x, gathered "in" := dy
            for li, l in enumerate(p['dg']):
                g = self._geom(p, 'wt', li, l, xv.N, dyv.H, dyv.W, self.Cop, dyv.ld, xv.H, xv.W, self.Cip, xv.ld)
                ex = 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co
                _timed('conv_wgrad', ex * self.algo_scale,
                       lambda: self._under_wg_share(lambda: nv.call(entry, g, rt.code, dyv.ptr, xv.ptr, nv.ptr(p['dwp'][li]),
                                                                    p['nslab'][li], rt.stream)),
                       ex)
                if unpack:
                    nv.call('segnb_unpack_wgrad', nv.ptr(p['dwp'][li]), nv.ptr(gw), self.Cip, self.Cop, len(l.taps),
                            self.s_in, self.s_out, p['tapoff_dg'][li], nv.ptr(self.in_map), nv.ptr(self.out_map), 1,
                            rt.stream)
            return
        for li, l in enumerate(p['fwd']):
            g = self._geom(p, 'f', li, l, xv.N, xv.H, xv.W, self.Cip, xv.ld, dyv.H, dyv.W, self.Cop, dyv.ld)
            _timed('conv_wgrad', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                   lambda: self._under_wg_share(lambda: nv.call(entry, g, rt.code, xv.ptr, dyv.ptr, nv.ptr(p['dwp'][li]),
                                                                p['nslab'][li], rt.stream)))
            if unpack:
                nv.call('segnb_unpack_wgrad', nv.ptr(p['dwp'][li]), nv.ptr(gw), self.Cop, self.Cip, len(l.taps),
                        self.s_out, self.s_in, p['tapoff_fwd'][li], nv.ptr(self.out_map), nv.ptr(self.in_map), 1,
                        rt.stream)


class UpConvOp(ConvOp):
    """The nearest-x2 upsampled segment of a convolution's input, computed on the LOW-resolution tensor.

    conv3x3(pad 1) over Upsample(scale_factor=2)(u)  (lib/models/zf_unet.py:42,78-90) is exactly the transposed
    convolution ConvTranspose2d(k=4, stride=2, pad=1)(u) whose 4x4 kernel is Wt[k] = sum of the 3x3 rows T[k]:
        k = 0: {dy = +1}    k = 1: {dy = 0, +1}    k = 2: {dy = -1, 0}    k = 3: {dy = -1}     (same along the width)
    i.e. per output parity 2 x 2 taps on u instead of 3 x 3 taps on the 4x larger upsampled tensor: 4/9 of the
    multiply-adds, forward, data gradient and weight gradient alike, and the upsampled copy is never read.  The op reuses
    ConvOp's transposed-convolution plan (phase launches forward, one 16-tap stride-2 gather for the data gradient, the
    'wt' weight-gradient geometry); only the weight pack / gradient unpack differ: they work on the reference's own 3x3
    parameter through MASKED jobs (PackJob.masked: one packed tap = the sum of several kernel positions).
    """
    T = ((2,), (1, 2), (0, 1), (0,))
    algo_scale = 9.0 / 4.0

    def __init__(self, rt, weight, ci_real, ci_pad, need_dgrad=True, pack_fwd=True):
        co, ci_total, kh, kw = weight.shape
        assert (kh, kw) == (3, 3) and ci_real <= ci_total and ci_pad % 8 == 0 and ci_pad >= ci_real
        self.rt, self.weight, self.bias = rt, weight, None
        self.out_hw_override = None
        self.stride, self.pad, self.transposed = 2, 1, True
        self.need_dgrad, self.pack_fwd = need_dgrad, pack_fwd
        self.Ci, self.Co, self.KH, self.KW = ci_real, co, 4, 4
        # element strides of the 3x3 parameter [Co][Ci_total][3][3], in ConvOp's transposed-convolution naming:
        # s_out = stride of one OUTPUT channel, s_in = stride of one input channel
        self.s_out, self.s_in = ci_total * 9, 9
        self.Cop, self.Cip = cp.pad8(co), ci_pad
        self.in_map = rt.int32(list(range(ci_real)) + [-1] * (ci_pad - ci_real))
        self.out_map = rt.int32(list(range(co)) + [-1] * (self.Cop - co))
        self._plans = {}

    def plan(self, Hi, Wi):
        p = ConvOp.plan(self, Hi, Wi)
        if 'wp_fwd_all' not in p:
            # the four phase matrices in ONE buffer [4][Cop][4 * Cip] (segnb_upconv_fprop_acc runs the phases as one launch)
            p['wp_fwd_all'] = self.rt.zeros((len(p['fwd']), self.Cop, 4 * self.Cip))
            p['wp_fwd'] = [p['wp_fwd_all'][i] for i in range(len(p['fwd']))]
        return p

    def fprop_acc_ok(self, uv, yv):
        return bool(nv.query('segnb_upconv_fprop_acc_ok', uv.N, uv.H, uv.W, self.Cip, self.Cop, yv.ld, self.rt.code))

    def fprop_acc(self, uv, yv, stats=None):
        """yv += conv3x3(Upsample x2(uv)) on the low-resolution tensor, the four output phases in one launch; stats: the
        BatchNorm statistics of the sums.  (Check fprop_acc_ok first.)"""
        p, rt = self.plan(uv.H, uv.W), self.rt
        assert uv.Cp == self.Cip and yv.Cp == self.Cop and (yv.H, yv.W) == (2 * uv.H, 2 * uv.W)
        ex = 2.0 * uv.N * uv.H * uv.W * 16 * self.Ci * self.Co
        _timed('conv_fprop', ex * self.algo_scale,
               lambda: nv.call('segnb_upconv_fprop_acc', rt.code, uv.N, uv.H, uv.W, self.Cip, uv.ld, uv.ptr,
                               nv.ptr(p['wp_fwd_all']), self.Cop, self.Cop, yv.ptr, yv.ld, nv.ptr(stats), rt.stream), ex)

    @classmethod
    def mask(cls, a, b):
        """bit mask over the nine 3x3 positions (kh * 3 + kw) summed into position (a, b) of the 4x4 kernel"""
        return sum(1 << (kh * 3 + kw) for kh in cls.T[a] for kw in cls.T[b])

    def pack(self, Hi, Wi):
        raise NotImplementedError('UpConvOp packs through the batched job table only (masked jobs)')

    def pack_jobs(self, Hi, Wi):
        p, rt = self.plan(Hi, Wi), self.rt
        w = self.weight.detach()
        jobs = []
        for li, l in enumerate(p['fwd'] if self.pack_fwd else ()):
            # ConvOp's transposed naming: the forward matrix is [Co][taps][Ci]
            jobs.append(dict(w=w, packed=p['wp_fwd'][li], mmap=self.out_map, cmap=self.in_map, s_m=self.s_out,
                             s_c=self.s_in, Mp=self.Cop, Cp=self.Cip, ntaps=len(l.taps), dtype=rt.code, masked=True, form='f',
                             tap_off=[self.mask(a, b) for (_, _, a, b) in l.taps]))
        if self.need_dgrad:
            for li, l in enumerate(p['dg']):
                jobs.append(dict(w=w, packed=p['wp_dg'][li], mmap=self.in_map, cmap=self.out_map, s_m=self.s_in,
                                 s_c=self.s_out, Mp=self.Cip, Cp=self.Cop, ntaps=len(l.taps), dtype=rt.code, masked=True, form='d',
                                 tap_off=[self.mask(a, b) for (_, _, a, b) in l.taps]))
        return jobs

    def unpack_jobs(self, Hi, Wi, grad_w):
        p = self.plan(Hi, Wi)
        return [dict(w=grad_w, packed=p['dwp'][li], mmap=self.in_map, cmap=self.out_map, s_m=self.s_in, s_c=self.s_out,
                     Mp=self.Cip, Cp=self.Cop, ntaps=len(l.taps), dtype=nv.F32, masked=True,
                     nslab=1,
                     tap_off=[self.mask(a, b) for (_, _, a, b) in l.taps]) for li, l in enumerate(p['dg'])]


class UpCatConvOp(object):
    """conv3x3(cat([Upsample x2(u), skip])) -- the first convolution of every ZF_UNET decoder block
    (lib/models/zf_unet.py:78-90) -- with its BACKWARD split by input segment:
        skip segment : the ordinary 3x3 data / weight gradient restricted to the skip channels (same kernels, a row
                       range of the packed matrix, a channel slice of the views);
        up segment   : UpConvOp -- the 4x4 / stride-2 gather from dy straight to the LOW-resolution gradient of u and the
                       matching weight gradient: 16 instead of 36 multiply-adds per low-resolution pixel, and the
                       upsampled gradient slice (4x the size it is consumed at) is neither written nor read.
    The forward stays the one 9-tap launch over the concat buffer.  Presents ConvOp's interface to Stage; the plan binds
    the low-resolution views with bind_up() before backward."""

    # segment_fwd = True (class attribute): the forward by segment too (default: the one 9-tap launch over the concat buffer -- measured alone,
    # tools/upcat_bench.py: 91 -> 78, 96 -> 90, 88 -> 94 us at the 28x28 / 56x56 / 112x112 decoder levels: the 2 x 2-window tiles
    # are short (8-32 steps) and pay the tile epilogue as often; the step time does not move while the weight gradient still
    # reads the upsampled copy)
    segment_fwd = False
    # segment_wgrad = True (class attribute): the weight gradient by segment too (default: the one 9-tap launch over the concat buffer --
    # measured alone at the five decoder shapes of the timed configuration, tools/upcat_bench.py: the weight-gradient kernel's
    # fixed cost per launch (partial slabs + their reduction) eats the 4/9 of the upsampled segment's multiply-adds)
    segment_wgrad = False

    def __init__(self, rt, weight, bias, in_segments, need_dgrad=True):
        (up_real, up_pad), (sk_real, sk_pad) = in_segments
        self.rt, self.weight, self.bias = rt, weight, bias
        # (full: also the PLAIN data gradient, for the input sizes whose low-resolution gather has no fast kernel)
        self.full = ConvOp(rt, weight, bias, in_segments, 1, 1, False, need_dgrad=need_dgrad)
        self.skip = ConvOp(rt, weight, bias, [(sk_real, sk_pad)], 1, 1, False, need_dgrad=True, ci_offset=up_real)
        self.up = UpConvOp(rt, weight, up_real, up_pad, need_dgrad=True, pack_fwd=True)
        self.needs_u = self.segment_wgrad        # the plan keeps the low-resolution tensor only for the segmented weight gradient
        self.up_pad, self.sk_pad = up_pad, sk_pad
        self.Co, self.Cop, self.Ci, self.Cip = self.full.Co, self.full.Cop, self.full.Ci, self.full.Cip
        self.need_dgrad = need_dgrad
        self.stride, self.pad, self.transposed = 1, 1, False
        self._u = self._du = None
        self._seg = {}

    def bind_up(self, u, du):
        """u: the low-resolution activated tensor the up segment was upsampled from; du: receives its gradient"""
        self._u, self._du = u, du

    def segmented(self, N, H, W):
        """Is the data gradient of this convolution at input size H x W computed by segment?  Only where the
        low-resolution gather runs on a fast kernel (segnb_conv_fprop_upd_ok): on the general gather kernel it costs more
        than the 5/9 of the multiply-adds it saves (measured in situ: 198 / 103 us against 118 / 124 us plain at the
        224x224 / 14x14 decoder levels of the timed configuration)."""
        if not self.need_dgrad or self.upsum(N, H, W):
            return False
        key = (N, H, W)
        v = self._seg.get(key)
        if v is None:
            p = self.up.plan(H // 2, W // 2)
            g = self.up._geom(p, 'd', 0, p['dg'][0], N, H, W, self.up.Cop, self.up.Cop, H // 2, W // 2, self.up.Cip, self.up.Cip)
            v = self._seg[key] = (self.force_segmented or (self.force_thin and self.Cop <= 32 and W >= 64) or
                                  bool(nv.query('segnb_conv_fprop_upd_ok', g, self.rt.code)))
        return v

    # fused_upsum = False (class attribute): no fused Upsample backward in the thin level's data gradient (the segmented form, or the plain one, instead)
    fused_upsum = True

    def upsum(self, N, H, W):
        """Is the data gradient at input size H x W the PLAIN 9-tap launch with the Upsample(x2) backward fused into its store
        pass (segnb_conv_fprop_upsum)?  The thin, HBM-bound 224x224 level: all 36 multiply-adds per low-resolution pixel are
        done -- they do not bound the kernel -- and the 4x-sized gradient slice is still neither written nor read."""
        if not self.need_dgrad or not self.fused_upsum or self.force_segmented:
            return False
        key = ('s', N, H, W)
        v = self._seg.get(key)
        if v is None:
            p = self.full.plan(H, W)
            v = False
            if len(p['dg']) == 1 and p['dg_full']:
                g = self.full._geom(p, 'd', 0, p['dg'][0], N, H, W, self.Cop, self.Cop, H, W, self.Cip, self.Cip)
                v = bool(nv.query('segnb_conv_fprop_upsum_ok', g, self.rt.code, self.up_pad))
            self._seg[key] = v
        return v

    def writes_du(self, N, H, W):
        """Does the data gradient at this size write the gradient of the low-resolution tensor itself (bind_up's du)?"""
        return self.upsum(N, H, W) or self.segmented(N, H, W)

    def fwd_segmented(self, N, H, W, ld_out=None):
        """Is the forward at input size H x W computed by segment -- the skip segment's 9-tap launch, then the upsampled
        segment on the low-resolution tensor accumulating into its output (segnb_upconv_fprop_acc)?"""
        if not self.segment_fwd:
            return False
        key = ('f', N, H, W)
        v = self._seg.get(key)
        if v is None:
            ld = self.Cop if ld_out is None else ld_out
            v = self._seg[key] = bool(nv.query('segnb_upconv_fprop_acc_ok', N, H // 2, W // 2, self.up.Cip, self.up.Cop, ld,
                                               self.rt.code))
        return v

    # virtual_concat = False (class attribute): the upsampled copy is materialised in the concat buffer (round-2 data flow)
    virtual_concat = True

    def virtual(self, N, H, W):
        """VIRTUAL CONCAT at this size: the forward and the weight gradient (9 taps over all input channels, as before) read
        the upsampled segment from the LOW-resolution tensor -- the kernels' tile fetch maps pixel (h, w) to u pixel
        (h >> 1, w >> 1) (segnb_conv_fprop_upcat / segnb_conv_wgrad_upcat) -- so the upsampled copy, 4x the size of u, is
        neither written (by the BatchNorm pass of the block below) nor read."""
        if not self.virtual_concat:
            return False
        key = ('v', N, H, W)
        v = self._seg.get(key)
        if v is None:
            p = self.full.plan(H, W)
            g = self.full._geom(p, 'f', 0, p['fwd'][0], N, H, W, self.full.Cip, self.full.Cip, H, W, self.full.Cop, self.full.Cop)
            v = self._seg[key] = bool(nv.query('segnb_conv_upcat_ok', g, self.rt.code, self.up_pad))
        return v

    def reads_upsampled(self, N, H, W):
        """Does anything still read the upsampled copy in the concat buffer at this size?  (the plain 9-tap forward or weight
        gradient)"""
        if self.virtual(N, H, W):
            return False
        return not self.fwd_segmented(N, H, W) or not self.segment_wgrad

    def needs_low_res(self, N, H, W):
        """Does anything read the low-resolution tensor itself?  (virtual concat, segmented forward or weight gradient)"""
        return self.virtual(N, H, W) or self.fwd_segmented(N, H, W) or self.segment_wgrad

    def _upcat_args(self, xv):
        """(geometry of the whole 9-tap convolution over the logical concat, skip view, segnb_upcat_src)"""
        p = self.full.plan(xv.H, xv.W)
        l = p['fwd'][0]
        sk = xv.slice(self.up_pad, self.sk_pad)
        src = nv.UpcatSrc(self._u.ptr, self.up_pad, self._u.ld)
        return p, l, sk, src

    _seg = None
    # thin output (<= 32 channels, the 224x224 level): the layer is HBM-bound on the 4x-sized upsampled gradient slice, and
    # not writing / re-reading it pays even though the low-resolution gather then runs on the general kernel (A/B below)
    force_thin = True
    force_segmented = os.environ.get('SEGNB_SUBPIXEL', 'auto') == 'force'

    # ---- forward: the whole 9-tap convolution over the concat buffer
    def plan(self, Hi, Wi):
        return self.full.plan(Hi, Wi)

    def out_hw(self, Hi, Wi):
        return self.full.out_hw(Hi, Wi)

    def fprop(self, xv, yv, stats=None, epilogue=None):
        if epilogue is None and self._u is not None and self.virtual(xv.N, xv.H, xv.W):
            p, l, sk, src = self._upcat_args(xv)
            g = self.full._geom(p, 'fv', 0, l, xv.N, xv.H, xv.W, self.full.Cip, sk.ld, yv.H, yv.W, self.full.Cop, yv.ld)
            b = self.bias.detach() if self.bias is not None else None
            _timed('conv_fprop', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                   lambda: nv.call('segnb_conv_fprop_upcat', g, self.rt.code, sk.ptr, src, nv.ptr(p['wp_fwd'][0]), nv.ptr(b),
                                   self.Co if b is not None else 0, yv.ptr, nv.ptr(stats), self.rt.stream))
            return
        if epilogue is None and self._u is not None and self.fwd_segmented(xv.N, xv.H, xv.W, yv.ld):
            # skip segment (with the bias) -> yv, then the upsampled segment added on the low-resolution tensor; the
            # BatchNorm statistics are taken by the second launch, on the sums
            self.skip.fprop(xv.slice(self.up_pad, self.sk_pad), yv, None)
            self.up.fprop_acc(self._u, yv, stats)
            return
        return self.full.fprop(xv, yv, stats, epilogue)

    def act_epilogue_ok(self, H, W, *a):
        return False        # (inference keeps the separate activation pass for these five layers)

    def u8_direct_ok(self, *a):
        return False

    def dgrad_bnreduce_ok(self, dyv, dxv):
        return False

    def dgrad_actmask_ok(self, dyv, dxv):
        return False

    def dgrad_bnapply_ok(self, *a):
        return False

    # ---- backward, by segment
    def dgrad(self, dyv, dxv, bn_reduce=None):
        assert bn_reduce is None
        if self.upsum(dxv.N, dxv.H, dxv.W):
            assert self._du is not None
            f, rt = self.full, self.rt
            p = f.plan(dxv.H, dxv.W)
            l = p['dg'][0]
            g = f._geom(p, 'd', 0, l, dyv.N, dyv.H, dyv.W, self.Cop, dyv.ld, dxv.H, dxv.W, self.Cip, dxv.ld)
            dst = nv.UpcatSrc(self._du.ptr, self.up_pad, self._du.ld)
            _timed('conv_fprop', 2.0 * dyv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                   lambda: nv.call('segnb_conv_fprop_upsum', g, rt.code, dyv.ptr, nv.ptr(p['wp_dg'][0]), dxv.ptr, dst, rt.stream))
            return
        if not self.segmented(dxv.N, dxv.H, dxv.W):
            return self.full.dgrad(dyv, dxv)
        assert self._du is not None
        self.skip.dgrad(dyv, dxv.slice(self.up_pad, self.sk_pad))
        self.up.dgrad(dyv, self._du)

    def wgrad(self, xv, dyv, grad_w, unpack=True):
        if not self.segment_wgrad and self._u is not None and self.virtual(xv.N, xv.H, xv.W):
            assert not unpack
            p, l, sk, src = self._upcat_args(xv)
            g = self.full._geom(p, 'fv', 0, l, xv.N, xv.H, xv.W, self.full.Cip, sk.ld, dyv.H, dyv.W, self.full.Cop, dyv.ld)

            def launch():
                if self.full.direct_ok():
                    self.full._arm_target(p, 0, grad_w)
                nv.call('segnb_conv_wgrad_upcat', g, self.rt.code, sk.ptr, src, dyv.ptr, nv.ptr(p['dwp'][0]), p['nslab'][0],
                        self.rt.stream)
            _timed('conv_wgrad', 2.0 * xv.N * l.QH * l.QW * len(l.taps) * self.Ci * self.Co,
                   lambda: self.full._under_wg_share(launch))
            return
        if not self.segment_wgrad:
            return self.full.wgrad(xv, dyv, grad_w, unpack)
        assert not unpack and self._u is not None, 'segmented weight gradients are unpacked by the batched table'
        self.skip.wgrad(xv.slice(self.up_pad, self.sk_pad), dyv, grad_w, unpack=False)
        self.up.wgrad(self._u, dyv, grad_w, unpack=False)

    # ---- weight pack / gradient unpack jobs of all three ops (H, W: the convolution's own, high, resolution)
    def pack_jobs(self, H, W, N=None):
        """N: the batch size the plan runs at (the fast-kernel query is per geometry); None = segmented wherever possible"""
        dseg = self.segmented(N, H, W) if N is not None else True
        fseg = (self.fwd_segmented(N, H, W) and not self.virtual(N, H, W)) if N is not None else self.segment_fwd
        full, skip, up = self.full.pack_jobs(H, W), self.skip.pack_jobs(H, W), self.up.pack_jobs(H // 2, W // 2)

        def pick(jobs, op, fwd):
            return [j for j in jobs if (j['mmap'] is op.out_map) == fwd]
        jobs = (pick(skip, self.skip, True) + pick(up, self.up, True)) if fseg else pick(full, self.full, True)
        if self.need_dgrad:
            jobs += (pick(skip, self.skip, False) + pick(up, self.up, False)) if dseg else pick(full, self.full, False)
        return jobs

    def unpack_jobs(self, H, W, grad_w):
        if not self.segment_wgrad:
            return self.full.unpack_jobs(H, W, grad_w)
        return self.skip.unpack_jobs(H, W, grad_w) + self.up.unpack_jobs(H // 2, W // 2, grad_w)


PACK_JOB_DTYPE = np.dtype([('w', '<u8'), ('packed', '<u8'), ('mmap', '<u8'), ('cmap', '<u8'), ('s_m', '<i8'),
                           ('s_c', '<i8'), ('Mp', '<i4'), ('Cp', '<i4'), ('ntaps', '<i4'), ('dtype', '<i4'),
                           ('block_start', '<i4'), ('nslab', '<i4'), ('masked', '<i4'), ('pad_', '<i4'),
                           ('tap_off', '<i4', (nv.MAX_TAPS,))])


PACK_PAIR_DTYPE = np.dtype([('w', '<u8'), ('pf', '<u8'), ('pd', '<u8'), ('Ci', '<i4'), ('Co', '<i4'), ('Cip', '<i4'),
                            ('Cop', '<i4'), ('block_start', '<i4'), ('pad_', '<i4'), ('tapf', '<i4', (9,)), ('tapd', '<i4', (9,))])


def _identity_prefix(m):
    """n when the channel map is 0 .. n-1 followed by padding (-1) only, else None"""
    v = m.detach().cpu().tolist()
    n = sum(1 for x in v if x >= 0)
    return n if v[:n] == list(range(n)) else None


class PackTable(object):
    """Device-side job table for segnb_pack_weight_multi / segnb_unpack_wgrad_multi: every weight matrix of a
    model packed (or every gradient unpacked) by ONE launch.  jobs: dicts with the fields of PACK_JOB_DTYPE
    (tensors for the pointer fields)."""

    # elem_multi = False (class attribute): the jobs the tiled kernel refuses as one ctypes call + launch each (A/B)
    elem_multi = True

    def __init__(self, rt, jobs, entry, single_entry, defer=None):
        """defer: predicate over the jobs that stay with the single-form kernels; those it selects are NOT put in this table but
        returned in self.deferred (the caller runs them from a table of their own, e.g. on another stream)"""
        assert PACK_JOB_DTYPE.itemsize == nv.query('segnb_pack_job_bytes'), 'PackJob layout drifted from the ABI'
        rows, self.singles, self._keep = [], [], []
        erows, eblocks = [], 0               # element-wise batched table (parameter tensors wider than 3 x 3)
        blocks = 0
        # the masked jobs (UpConvOp: a packed tap sums several kernel positions) have few, slow blocks: FIRST in the table, their
        # blocks start with the launch and end inside it -- at the end of the table they were a 14 us tail of ZF_UNET's 102 us pack
        # (tools/pack_bench.py: 3 jobs, 336 of 21168 blocks)
        jobs = sorted(jobs, key=lambda j: 0 if j.get('masked') else 1)
        self.ptable, self.pn, self.pblocks, self.paired_ids = None, 0, 0, set()
        if entry == 'segnb_pack_weight_multi' and self.pair_pack:
            jobs = self._pair(rt, jobs)
        self.deferred = [j for j in jobs if defer(j)] if defer is not None else []
        if self.deferred:
            jobs = [j for j in jobs if not defer(j)]
        for j in jobs:
            nb = nv.query('segnb_pack_job_blocks', j['Mp'], j['Cp'], j['ntaps'], j['s_m'], j['s_c'])
            if nb < 0:                       # kernels wider than 3x3: the element-wise kernels, still ONE launch for all of them
                enb = nv.query('segnb_pack_elem_job_blocks', j['Mp'], j['Cp'], j['ntaps']) if self.elem_multi else -1
                if enb < 0 or j.get('masked') or j.get('nslab', 1) > 1:
                    self.singles.append(j)
                    continue
                row = np.zeros((), dtype=PACK_JOB_DTYPE)
                for f in ('w', 'packed', 'mmap', 'cmap'):
                    row[f] = j[f].data_ptr()
                    self._keep.append(j[f])
                for f in ('s_m', 's_c', 'Mp', 'Cp', 'ntaps', 'dtype'):
                    row[f] = j[f]
                row['nslab'] = 1
                row['tap_off'][:len(j['tap_off'])] = j['tap_off']
                row['block_start'] = eblocks
                eblocks += enb
                erows.append(row)
                continue
            row = np.zeros((), dtype=PACK_JOB_DTYPE)
            for f in ('w', 'packed', 'mmap', 'cmap'):
                row[f] = j[f].data_ptr()
                self._keep.append(j[f])
            for f in ('s_m', 's_c', 'Mp', 'Cp', 'ntaps', 'dtype'):
                row[f] = j[f]
            row['nslab'] = j.get('nslab', 1)
            row['masked'] = 1 if j.get('masked') else 0
            row['tap_off'][:len(j['tap_off'])] = j['tap_off']
            row['block_start'] = blocks
            blocks += nb
            rows.append(row)
        self.rt, self.entry, self.single_entry, self.n, self.blocks = rt, entry, single_entry, len(rows), blocks
        self.table = None
        if rows:
            tab = np.array(rows, dtype=PACK_JOB_DTYPE)
            self.table = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(rt.device)
        self.etable, self.en, self.eblocks = None, len(erows), eblocks
        self.eentry = 'segnb_pack_weight_elem_multi' if entry == 'segnb_pack_weight_multi' else 'segnb_unpack_wgrad_elem_multi'
        if erows:
            tab = np.array(erows, dtype=PACK_JOB_DTYPE)
            self.etable = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(rt.device)

    # pair_pack = False (class attribute): every matrix by its own job (A/B)
    pair_pack = True

    def _pair(self, rt, jobs):
        """The forward and the data-gradient matrix of a plain 3x3 convolution (ConvOp.pack_jobs: form 'f' / 'd' on the same
        parameter, channels in place) as ONE job of segnb_pack_weight_pair_multi: the parameter is read once.  Returns the jobs
        that stay with the single-form kernels."""
        by_w = {}
        for j in jobs:
            if (j.get('form') in ('f', 'd') and not j.get('masked') and j['ntaps'] == 9 and j['dtype'] == nv.BF16
                    and j.get('nslab', 1) == 1 and sorted(j['tap_off']) == list(range(9))):
                by_w.setdefault(j['w'].data_ptr(), []).append(j)
        rows, taken, blocks = [], set(), 0
        for js in by_w.values():
            fs = [j for j in js if j['form'] == 'f']
            if len(fs) != 1:
                continue
            jf = fs[0]
            ci, co = _identity_prefix(jf['cmap']), _identity_prefix(jf['mmap'])
            if ci is None or co is None or jf['s_c'] != 9 or jf['s_m'] != 9 * ci or jf['w'].numel() != co * ci * 9:
                continue
            # the data-gradient matrix of the same parameter, channels in place too (a segmented data gradient -- UpCatConvOp --
            # keeps its own jobs: the forward matrix is then packed alone)
            ds = [j for j in js if j['form'] == 'd' and j['s_m'] == 9 and j['s_c'] == 9 * ci and j['Mp'] == jf['Cp']
                  and j['Cp'] == jf['Mp'] and j['mmap'] is jf['cmap'] and j['cmap'] is jf['mmap']]
            jd = ds[0] if len(ds) == 1 else None
            nb = nv.query('segnb_pack_pair_job_blocks', co, ci, jf['Mp'], jf['Cp'])
            if nb < 0:
                continue
            row = np.zeros((), dtype=PACK_PAIR_DTYPE)
            row['w'], row['pf'] = jf['w'].data_ptr(), jf['packed'].data_ptr()
            row['pd'] = jd['packed'].data_ptr() if jd is not None else 0
            self._keep += [jf['w'], jf['packed']] + ([jd['packed']] if jd is not None else [])
            row['Ci'], row['Co'], row['Cip'], row['Cop'] = ci, co, jf['Cp'], jf['Mp']
            row['tapf'][:] = jf['tap_off']
            row['tapd'][:] = jd['tap_off'] if jd is not None else jf['tap_off']
            row['block_start'] = blocks
            blocks += nb
            rows.append(row)
            taken.add(id(jf))
            if jd is not None:
                taken.add(id(jd))
        if rows:
            assert PACK_PAIR_DTYPE.itemsize == nv.query('segnb_pack_pair_job_bytes'), 'PackPairJob layout drifted from the ABI'
            tab = np.array(rows, dtype=PACK_PAIR_DTYPE)
            self.ptable = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(rt.device)
            self.pn, self.pblocks = len(rows), blocks
            self.pair_params = [(int(r['w']), int(r['Co']) * int(r['Ci']) * 9) for r in rows]      # (address, elements) of every paired parameter
        self.paired_ids = taken
        return [j for j in jobs if id(j) not in taken]

    pair_params = ()

    def run_pairs_sgd(self, flat_p, flat_g, lr):
        """optimizer.step() of plain SGD on the paired parameters and their pack in one pass (segnb_sgd_pack_pair_multi); the rest of
        the table (run(skip_pairs=True)) is packed at the next forward as usual"""
        nv.call('segnb_sgd_pack_pair_multi', nv.ptr(self.ptable), self.pn, self.pblocks, nv.ptr(flat_p), nv.ptr(flat_g), float(lr),
                self.rt.stream)

    def run(self, skip_pairs=False):
        if self.ptable is not None and not skip_pairs:
            nv.call('segnb_pack_weight_pair_multi', nv.ptr(self.ptable), self.pn, self.pblocks, self.rt.stream)
        if self.table is not None:
            nv.call(self.entry, nv.ptr(self.table), self.n, self.blocks, self.rt.stream)
        if self.etable is not None:
            nv.call(self.eentry, nv.ptr(self.etable), self.en, self.eblocks, self.rt.stream)
        for j in self.singles:
            assert not j.get('masked'), 'masked jobs exist in the batched form only'
            tap = nv.int_array(j['tap_off'])
            if self.single_entry == 'segnb_pack_weight':
                nv.call('segnb_pack_weight', nv.ptr(j['w']), nv.ptr(j['packed']), j['dtype'], j['Mp'], j['Cp'],
                        j['ntaps'], j['s_m'], j['s_c'], tap, nv.ptr(j['mmap']), nv.ptr(j['cmap']), self.rt.stream)
            else:
                nv.call('segnb_unpack_wgrad', nv.ptr(j['packed']), nv.ptr(j['w']), j['Mp'], j['Cp'], j['ntaps'],
                        j['s_m'], j['s_c'], tap, nv.ptr(j['mmap']), nv.ptr(j['cmap']), 1, self.rt.stream)


class Stage(object):
    """conv -> [BatchNorm2d] -> activation -> [Dropout2d multiplier] with optional fused MaxPool2d(2) /
    nearest-x2 outputs.  (_Conv3BN of lib/models/zf_unet.py:5-17 plus the Dropout2d/pool/unpool that
    follow it at :31,:41,:42.)"""

    direct_apply = True
    # Layers whose gradient has several sources, a pooled source or a Dropout2d multiplier: dz need not be stored either -- the
    # apply pass re-reads the sources and recomputes it (segnb_bn_bwd_apply_fused_src: one tensor write and one read less per
    # layer) for tensors of at least this many MB.  OFF (0) by default: measured on MI355X at 32 MB (the 224 x 224 / 112 x 112
    # levels), same box, alternating runs, 5.36 / 5.33 ms per step with it against 5.29 / 5.30 without -- the pooled-window walk
    # of the source pass runs at 3 TB/s where the plain apply pass it replaces runs at 6.4 (profiles/r04_ab.txt).
    recompute_dz_min_mb = 0.0

    def __init__(self, rt, conv, bn=None, act=nv.ACT_RELU, slope=0.01, name=''):
        self.rt, self.conv, self.bn, self.act, self.slope, self.name = rt, conv, bn, act, slope, name
        self.defer_unpack = False     # True: the model plan runs one batched unpack at the end of backward
        # BatchNorm finalize folded into the activation / apply passes (segnb_bn_fwd_fused / _bwd_apply_fused): saves
        # two 5 us launches per layer, but every block of the big kernel then starts with the same dependent
        # statistics loads -- measured neutral on MI355X (7.47 vs 7.39 ms/step), so off by default
        # finalize folded into the activation / apply passes: 44 launches of ~5 us less on the dependent chain
        # (re-measured after the convolutions got faster: 5.71 -> 5.58 ms/step; neutral when first tried)
        self.fuse_finalize = os.environ.get('SEGNB_FUSE_FINALIZE', '1') != '0'
        self._stats_stale = False
        self._fused_fwd = False
        Cp = conv.Cop
        self.C, self.Cp = conv.Co, Cp
        self.stats = rt.zeros((STAT_REPLICAS, 2, Cp), torch.float64)   # consumed + re-zeroed by segnb_bn_finalize
        self.sums = rt.zeros((STAT_REPLICAS, 2, Cp), torch.float64)    # consumed + re-zeroed by segnb_bn_bwd_finalize
        self.coef = rt.zeros((4, Cp), torch.float32)
        self.bcoef = rt.zeros((3, Cp), torch.float32)
        self._bufs = {}

    def buffers(self, N, Ho, Wo):
        key = (N, Ho, Wo)
        b = self._bufs.get(key)
        if b is None:
            b = {'y': View.alloc(self.rt, N, Ho, Wo, self.Cp), 'dz': View.alloc(self.rt, N, Ho, Wo, self.Cp)}
            self._bufs[key] = b
        return b

    def defer_act_ok(self, xv, train, need_grad):
        """Can this stage skip its activation pass and leave BatchNorm + activation to the consumer's loads (defer_act)?"""
        return bool(self.bn is not None and train and need_grad and self.fuse_finalize and self.consumer_fusion)

    # BatchNorm + activation of a block's FIRST convolution applied by the second one while it loads its rows (conv_roll_kernel,
    # segnb_conv_fprop_tf): the activated tensor between them is never written (-0.4 GB of HBM traffic per ZF_UNET step at
    # bs=32).  ON since round 6 (SEGNB_CONSUMER_FUSION=0 restores the passes): the two activation passes it removes (2 x ~38 us) come
    # back as slower convolutions and weight gradients, whose loads carry ~10 VALU operations per element -- round 4 measured
    # -0.1 % (5.368 / 5.378 against 5.373 / 5.384), round 6 on three boxes -0.35 % (4.774 / 4.797 / 4.795 against 4.805 / 4.805 / 4.807;
    # 4.761 / 4.768 against 4.783 / 4.781: profiles/r06_ab.txt).  The teacher-forced replay that "failed" with it in round 5 compared
    # the 16 statistics replicas of segnb_conv_fprop_tf one by one instead of their sum (tests/abi_replay.py: _REPLICATED).
    consumer_fusion = os.environ.get('SEGNB_CONSUMER_FUSION', '1') != '0'

    def forward(self, xv, train, dropmul=None, out=None, pool_out=None, up_out=None, need_grad=True, u8=None, x_tf=None,
                defer_act=False):
        """u8: (uint8 NHWC batch, InputNorm) -- this stage is the network's first convolution and reads the image
        itself (segnb_conv_fprop_u8); xv then only RECEIVES the normalised pixels (for the weight gradient).
        x_tf: xv is the PRE-BatchNorm output of the producing stage and x_tf the segnb_operand_tf that turns it into this
        convolution's input (ConvOp.tf_act): forward and weight gradient apply it while they load.
        defer_act: no activation pass -- the caller hands (y View, tf) of this stage to its consumer (returned by tf_out())."""
        rt = self.rt
        Ho, Wo = self.conv.out_hw(xv.H, xv.W)
        b = self.buffers(xv.N, Ho, Wo)
        yv = b['y']
        use_batch_stats = self.bn is not None and train
        if use_batch_stats and self._stats_stale:
            # the previous training-mode forward was fused (statistics left unconsumed) and no backward cleared them
            self.stats.zero_()
            self._stats_stale = False
        if (not need_grad and not use_batch_stats and u8 is None and out is not None and pool_out is None and up_out is None
                and dropmul is None and self.conv.act_epilogue_ok(xv.H, xv.W)):
            # inference (validate(), tiled prediction): eval-mode BatchNorm and the activation in the convolution's epilogue --
            # the activated output goes straight to `out`, no BatchNorm pass at all
            coef = None
            if self.bn is not None:
                bn = self.bn
                nv.call('segnb_bn_finalize', nv.ptr(self.stats), self.C, self.Cp, float(xv.N * Ho * Wo),
                        nv.ptr(bn.weight.detach()), nv.ptr(bn.bias.detach()), BN_EPS, BN_MOMENTUM,
                        nv.ptr(bn.running_mean), nv.ptr(bn.running_var), nv.ptr(bn.num_batches_tracked), 0,
                        nv.ptr(self.coef), rt.stream)
                coef = self.coef
            self.conv.fprop(xv, out, None, epilogue=(coef, self.act, self.slope))
            self._saved = None
            return out
        self._x_tf = x_tf
        if u8 is not None:
            self.conv.fprop_u8(u8[0], u8[1], yv, self.stats if use_batch_stats else None, xv if need_grad else None)
        elif x_tf is not None:
            self.conv.fprop_tf(xv, x_tf, yv, self.stats if use_batch_stats else None)
        else:
            self.conv.fprop(xv, yv, self.stats if use_batch_stats else None)
        coef = None
        fused = use_batch_stats and self.fuse_finalize and need_grad
        if defer_act == 'head':
            # the network's last layer: its activation pass is launched by the plan together with the classifier behind it
            # (head_forward: segnb_bn_fwd_fused_head), outside any recorded list -- the logits go to a fresh tensor
            assert fused and pool_out is None and up_out is None
            self._stats_stale = True
            self._saved = (xv, yv, dropmul, True)
            self._fused_fwd = True
            return yv
        if defer_act:
            assert fused and dropmul is None and pool_out is None and up_out is None
            bn = self.bn
            nv.call('segnb_bn_finalize_keep', nv.ptr(self.stats), self.C, self.Cp, float(xv.N * Ho * Wo),
                    nv.ptr(bn.weight.detach()), nv.ptr(bn.bias.detach()), BN_EPS, BN_MOMENTUM, nv.ptr(bn.running_mean),
                    nv.ptr(bn.running_var), nv.ptr(bn.num_batches_tracked), nv.ptr(self.coef), nv.ptr(self.sums), rt.stream)
            self._stats_stale = True
            self._saved = (xv, yv, None, True)
            self._fused_fwd = True
            return yv
        if fused:
            # finalize folded into the activation pass (one launch less per layer and direction); the statistics are
            # cleared by this layer's backward (segnb_bn_bwd_apply_fused), the backward sums here
            bn = self.bn
            nv.call('segnb_bn_fwd_fused', rt.code, yv.ptr, yv.ld, xv.N, Ho, Wo, self.C, self.Cp, nv.ptr(self.stats),
                    nv.ptr(bn.weight.detach()), nv.ptr(bn.bias.detach()), BN_EPS, BN_MOMENTUM,
                    nv.ptr(bn.running_mean), nv.ptr(bn.running_var), nv.ptr(bn.num_batches_tracked),
                    nv.ptr(self.coef), nv.ptr(self.sums), self.act, self.slope, nv.ptr(dropmul), vptr(out), vld(out),
                    vptr(pool_out), vld(pool_out), vptr(up_out), vld(up_out), None, 0, rt.stream)
            self._stats_stale = True
            self._saved = (xv, yv, dropmul, True)
            self._fused_fwd = True
            return yv
        self._fused_fwd = False
        if self.bn is not None:
            bn = self.bn
            nv.call('segnb_bn_finalize', nv.ptr(self.stats), self.C, self.Cp, float(xv.N * Ho * Wo),
                    nv.ptr(bn.weight.detach()), nv.ptr(bn.bias.detach()), BN_EPS, BN_MOMENTUM,
                    nv.ptr(bn.running_mean), nv.ptr(bn.running_var), nv.ptr(bn.num_batches_tracked),
                    1 if train else 0, nv.ptr(self.coef), rt.stream)
            coef = self.coef
        nv.call('segnb_bn_act_fwd', rt.code, yv.ptr, yv.ld, xv.N, Ho, Wo, self.Cp, nv.ptr(coef), self.act,
                self.slope, nv.ptr(dropmul), vptr(out), vld(out), vptr(pool_out), vld(pool_out), vptr(up_out),
                vld(up_out), None, 0, rt.stream)
        self._saved = (xv, yv, dropmul, coef is not None)
        return yv

    def head_forward(self, head_w, head_b, K, logits, out=None):
        """BatchNorm (finalize folded in) + activation (+ Dropout2d) of this stage's convolution output AND the 1x1 classifier on
        the activated values, one launch (after forward(..., defer_act='head')); out: optional View for the activated tensor."""
        rt, bn = self.rt, self.bn
        xv, yv, dropmul, _ = self._saved
        nv.call('segnb_bn_fwd_fused_head', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.C, self.Cp, nv.ptr(self.stats),
                nv.ptr(bn.weight.detach()), nv.ptr(bn.bias.detach()), BN_EPS, BN_MOMENTUM, nv.ptr(bn.running_mean),
                nv.ptr(bn.running_var), nv.ptr(bn.num_batches_tracked), nv.ptr(self.coef), nv.ptr(self.sums), self.act,
                self.slope, nv.ptr(dropmul), vptr(out), vld(out), nv.ptr(head_w), nv.ptr(head_b), K, nv.ptr(logits), rt.stream)

    def head_backward(self, head_w, K, dlogits, dw, db):
        """d(logits) through the classifier, this stage's activation / Dropout2d and its BatchNorm-backward reduction in one pass
        over y (segnb_head_bn_bwd): leaves dz in the stage's buffer and the sums complete -- backward(..., dz_ready=True) next."""
        rt = self.rt
        xv, yv, dropmul, _ = self._saved
        dz = self.buffers(yv.N, yv.H, yv.W)['dz']
        # head_dz_recompute: dz is NOT stored -- backward(dz_ready=True) recomputes it from d(logits) and y inside the apply pass
        # (segnb_head_bn_bwd_apply): one tensor write and one tensor read less at the network's full resolution
        self._head_src = (head_w, K, dlogits) if (self.head_dz_recompute and self.bn is not None and self.fuse_finalize) else None
        nv.call('segnb_head_bn_bwd', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.C, self.Cp, nv.ptr(self.coef), self.act,
                self.slope, nv.ptr(dropmul), nv.ptr(head_w), K, nv.ptr(dlogits), None if self._head_src is not None else dz.ptr,
                dz.ld, nv.ptr(self.sums), nv.ptr(dw), nv.ptr(db), rt.stream)

    # head_dz_recompute = False (class attribute / SEGNB_HEAD_DZ=0): the last layer's dz is stored by segnb_head_bn_bwd and read back (A/B)
    head_dz_recompute = os.environ.get('SEGNB_HEAD_DZ', '1') != '0'
    _head_src = None

    def tf_out(self):
        """the operand transform a consumer applies to this stage's pre-BatchNorm output (after forward(..., defer_act=True))"""
        return ConvOp.tf_act(self.coef, self.Cp, self.act, self.slope)

    def reduce_in_producer(self):
        """-> the (y, coef, sums, act, slope) a data-gradient launch needs to do THIS layer's BatchNorm-backward
        reduction in its epilogue, or None when the layer does not qualify (it must be a 'direct' layer: BatchNorm,
        one direct gradient source, no dropout; the caller guarantees the single direct source)."""
        xv, yv, dropmul, has_bn = self._saved
        if not (self.direct_apply and has_bn and dropmul is None and self._fused_fwd):
            return None
        return (yv, self.coef, self.sums, self.act, self.slope)

    def backward(self, grads, g_direct=None, g_pool=None, g_up=None, dx=None, reduced=False, fuse_reduce_of=None,
                 dz_ready=False):
        """grads: FlatParams (gives the fp32 gradient view of each parameter).  dx: View to receive the
        input gradient, or None (first layer).  reduced: the reduction pass of this layer was already done by the
        data-gradient launch that produced g_direct.  fuse_reduce_of: the Stage whose activation gradient dx is -- if
        it qualifies, this layer's data gradient does that stage's reduction too; returns True when it did."""
        rt = self.rt
        xv, yv, dropmul, has_bn = self._saved
        dz = self.buffers(yv.N, yv.H, yv.W)['dz']
        coef = self.coef if has_bn else None
        # A single direct gradient source, no dropout: dz never goes to memory -- the reduce pass only sums, the apply
        # pass recomputes dz from g (segnb_bn_bwd_apply_direct): one tensor write less per such layer.
        # dz_ready: head_backward() already left dz in the buffer and completed the sums (no gradient source tensor at all)
        direct = (self.direct_apply and has_bn and g_direct is not None and g_pool is None and g_up is None
                  and dropmul is None and not dz_ready)
        assert not reduced or direct, 'only a direct layer can be reduced by its producer'
        mb = yv.N * yv.H * yv.W * self.Cp * (2 if rt.code == nv.BF16 else 4) / 1e6
        recompute = (not direct and not dz_ready and has_bn and self._fused_fwd and self.recompute_dz_min_mb > 0
                     and mb >= self.recompute_dz_min_mb)
        if not reduced and not dz_ready:
            nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.Cp, nv.ptr(coef),
                    self.act, self.slope, nv.ptr(dropmul), vptr(g_direct), vld(g_direct), vptr(g_pool), vld(g_pool),
                    vptr(g_up), vld(g_up), None if (direct or recompute) else dz.ptr, dz.ld, nv.ptr(self.sums), None, 0,
                    rt.stream)
        count = float(yv.N * yv.H * yv.W)
        gbias = grads.grad_of(self.conv.bias) if self.conv.bias is not None else None
        if (dx is None and direct and self._fused_fwd and self.defer_unpack
                and isinstance(self.conv, ConvOp) and self.conv.wgrad_bnapply_ok(xv, yv)):
            # FIRST layer of the network: no data gradient, so the only reader of dy is this layer's weight gradient -- it
            # recomputes dy from (g, y) while staging its tiles (segnb_conv_wgrad_bnapply): the apply pass and its tensor are
            # gone from the serial tail of backward (a 5 us finalize instead of a full pass over the largest activation)
            nv.call('segnb_bn_bwd_finalize_clear', nv.ptr(self.sums), self.C, self.Cp, count, nv.ptr(self.bn.weight.detach()),
                    nv.ptr(self.coef), nv.ptr(self.bcoef), nv.ptr(grads.grad_of(self.bn.weight)),
                    nv.ptr(grads.grad_of(self.bn.bias)), 1, nv.ptr(self.stats), rt.stream)
            self._stats_stale = False
            self.conv.wgrad_bnapply(xv, g_direct, yv, self.coef, self.bcoef, self.act, self.slope,
                                    grad_w=grads.grad_of(self.conv.weight))
            return False
        # the weight gradient is forked to the side stream right behind the apply pass: its event rides on that launch
        if self.defer_unpack and dx is not None and rt.side_stream() is not None:
            rt.arm_fork()
        if dz_ready and self._head_src is not None and has_bn and self._fused_fwd:
            head_w, K, dlogits = self._head_src
            nv.call('segnb_head_bn_bwd_apply', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.C, self.Cp, nv.ptr(self.coef),
                    nv.ptr(self.sums), nv.ptr(self.bn.weight.detach()), nv.ptr(self.bcoef), nv.ptr(grads.grad_of(self.bn.weight)),
                    nv.ptr(grads.grad_of(self.bn.bias)), 1, nv.ptr(self.stats), self.act, self.slope, nv.ptr(dropmul),
                    nv.ptr(head_w), K, nv.ptr(dlogits), dz.ptr, dz.ld, rt.stream)
            self._stats_stale = False
        elif has_bn and self._fused_fwd and direct:
            nv.call('segnb_bn_bwd_apply_fused_direct', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.C, self.Cp,
                    nv.ptr(self.coef), nv.ptr(self.sums), nv.ptr(self.bn.weight.detach()), nv.ptr(self.bcoef),
                    nv.ptr(grads.grad_of(self.bn.weight)), nv.ptr(grads.grad_of(self.bn.bias)), 1,
                    nv.ptr(self.stats), self.act, self.slope, g_direct.ptr, g_direct.ld, dz.ptr, dz.ld, rt.stream)
            self._stats_stale = False
        elif recompute:
            nv.call('segnb_bn_bwd_apply_fused_src', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.C, self.Cp,
                    nv.ptr(self.coef), nv.ptr(self.sums), nv.ptr(self.bn.weight.detach()), nv.ptr(self.bcoef),
                    nv.ptr(grads.grad_of(self.bn.weight)), nv.ptr(grads.grad_of(self.bn.bias)), 1, nv.ptr(self.stats),
                    self.act, self.slope, nv.ptr(dropmul), vptr(g_direct), vld(g_direct), vptr(g_pool), vld(g_pool),
                    vptr(g_up), vld(g_up), dz.ptr, dz.ld, rt.stream)
            self._stats_stale = False
        elif has_bn and self._fused_fwd:
            nv.call('segnb_bn_bwd_apply_fused', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.C, self.Cp,
                    nv.ptr(self.coef), nv.ptr(self.sums), nv.ptr(self.bn.weight.detach()), nv.ptr(self.bcoef),
                    nv.ptr(grads.grad_of(self.bn.weight)), nv.ptr(grads.grad_of(self.bn.bias)), 1,
                    nv.ptr(self.stats), dz.ptr, dz.ld, dz.ptr, dz.ld, rt.stream)
            self._stats_stale = False
        elif has_bn:
            nv.call('segnb_bn_bwd_finalize', nv.ptr(self.sums), self.C, self.Cp, count,
                    nv.ptr(self.bn.weight.detach()), nv.ptr(self.coef), nv.ptr(self.bcoef),
                    nv.ptr(grads.grad_of(self.bn.weight)), nv.ptr(grads.grad_of(self.bn.bias)), 1, rt.stream)
            # d(loss)/d(conv bias) under training-mode BatchNorm is identically zero (BN subtracts the batch mean):
            # sum(dy) = A*(sum dz - n*mean(dz) - mean(dz*yhat)*sum(yhat)) = 0.  The reference's fp32 value is pure
            # summation noise (~1e-7 of the weight-gradient scale); the flat gradient buffer already holds 0.
            if direct:
                nv.call('segnb_bn_bwd_apply_direct', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.Cp,
                        nv.ptr(self.coef), nv.ptr(self.bcoef), self.act, self.slope, g_direct.ptr, g_direct.ld,
                        dz.ptr, dz.ld, None, self.C, rt.stream)
            else:
                nv.call('segnb_bn_bwd_apply', rt.code, yv.ptr, yv.ld, yv.N, yv.H, yv.W, self.Cp, nv.ptr(self.coef),
                        nv.ptr(self.bcoef), dz.ptr, dz.ld, dz.ptr, dz.ld, None, self.C, rt.stream)
        else:
            # no BatchNorm: dy = dz, d(bias) = sum dz (accumulated through the dbeta slot)
            nv.call('segnb_bn_bwd_finalize', nv.ptr(self.sums), self.C, self.Cp, count, None, nv.ptr(self.coef),
                    nv.ptr(self.bcoef), None, nv.ptr(gbias), 1, rt.stream)
        x_tf = getattr(self, '_x_tf', None)

        def wgrad(unpack):
            if x_tf is not None:
                # the convolution's input is not in memory: recomputed from the producer's pre-BatchNorm output on load
                assert self.defer_unpack and not unpack
                self.conv.wgrad_tf(xv, x_tf, dz, None, grad_w=grads.grad_of(self.conv.weight))
            else:
                self.conv.wgrad(xv, dz, grads.grad_of(self.conv.weight), unpack=unpack)
        side = rt.fork_side() if (self.defer_unpack and dx is not None) else None
        if side is not None:
            with torch.cuda.stream(side):
                wgrad(False)
        else:
            wgrad(not self.defer_unpack)
        fused = False
        if dx is not None:
            ep = fuse_reduce_of.reduce_in_producer() if fuse_reduce_of is not None else None
            if ep is not None and self.conv.dgrad_bnreduce_ok(dz, dx):
                self.conv.dgrad(dz, dx, bn_reduce=ep)
                fused = True
            else:
                self.conv.dgrad(dz, dx)
        return fused


class FlatParams(object):
    """All parameters of a module as views of ONE fp32 buffer, all gradients as views of another
    (one SGD kernel, one all-reduce bucket list, one memset).  nn.Parameter identity is preserved, so
    torch.optim / state_dict / checkpoints see ordinary parameters (torch_train.py:375, :308-330)."""

    registry = {}      # id(parameter) -> FlatParams owning it (lets segnb.optim find the flat buffers)

    def __init__(self, module):
        self.module = module
        self.flat_p = None
        self.flat_g = None
        self._off = {}
        self._gviews = {}
        self._slots = None
        self.version = 0   # bumped by in-place updates that bypass torch's version counters (fused SGD)

    # module.parameters() / .buffers() walk the whole module tree (0.1-0.2 ms for a 60-module net, several times per step):
    # the walk is done once and remembered as (owning module, attribute name) slots, which still see a replaced Parameter
    # or buffer object; a net whose sub-modules are added or removed afterwards calls forget_structure()
    def _structure(self):
        if self._slots is None:
            ps, bs, seen = [], [], set()
            for m in self.module.modules():
                for n, p in m._parameters.items():
                    if p is not None and id(p) not in seen:
                        seen.add(id(p))
                        ps.append((m, n))
                for n, b in m._buffers.items():
                    if b is not None and id(b) not in seen:
                        seen.add(id(b))
                        bs.append((m, n))
            self._slots = (ps, bs)
        return self._slots

    def forget_structure(self):
        self._slots = None
        self._modules = None

    _modules = None

    def module_list(self):
        if self._modules is None:
            self._modules = list(self.module.modules())
        return self._modules

    def param_list(self):
        out = [m._parameters[n] for m, n in self._structure()[0]]
        if any(p is None for p in out):
            self._slots = None
            out = [m._parameters[n] for m, n in self._structure()[0]]
        return out

    def buffer_list(self):
        out = [m._buffers[n] for m, n in self._structure()[1]]
        if any(b is None for b in out):
            self._slots = None
            out = [m._buffers[n] for m, n in self._structure()[1]]
        return out

    def ensure(self, device):
        params = self.param_list()
        ok = self.flat_p is not None and self.flat_p.device == device
        if ok:
            base = self.flat_p.data_ptr()
            for p in params:
                off = self._off.get(id(p))
                if off is None or p.data_ptr() != base + 4 * off[0]:
                    ok = False
                    break
        if ok:
            return
        total, offs = 0, {}
        for p in params:
            if p.dtype != torch.float32:
                raise TypeError('parameters must be float32 (the reference trains in fp32)')
            n = p.numel()
            offs[id(p)] = (total, n)
            total += (n + 3) // 4 * 4          # keep every view 16-byte aligned
        flat_p = torch.zeros(total, dtype=torch.float32, device=device)
        flat_g = torch.zeros(total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p in params:
                off, n = offs[id(p)]
                view = flat_p[off:off + n].view(p.shape)
                view.copy_(p.data.to(device))
                p.data = view
        self.flat_p, self.flat_g, self._off = flat_p, flat_g, offs
        self._gviews = {id(p): flat_g[offs[id(p)][0]:offs[id(p)][0] + offs[id(p)][1]].view(p.shape) for p in params}
        self.total = total
        self.version += 1
        for p in params:
            FlatParams.registry[id(p)] = self

    # optimizer.step() of plain SGD fused with the next forward's weight pack (segnb.optim.SGD -> the model plan's hook):
    # sgd_pack_hook(lr) -> True when it applied the update to EVERY parameter (and packed what it could); sgd_pack_done() is called
    # after the optimizer bumped `version`
    sgd_pack_hook = None
    sgd_pack_done = None

    def grad_of(self, p):
        if self.touch_log is not None:
            self.touch_log.add(id(p))
        return self._gviews[id(p)]

    touch_log = None       # a set while a backward learns which closure finishes which parameters (segnb.net.Tape)

    def grad_absmax(self):
        """max |g| over every parameter gradient, as a 0-dim device tensor: ONE launch over the flat gradient buffer
        (the reference loops over named_parameters with a reduction and a host sync per tensor, torch_train.py:199-203)."""
        out = torch.zeros((), dtype=torch.float32, device=self.flat_g.device)
        st = torch.cuda.current_stream(self.flat_g.device).cuda_stream if self.flat_g.is_cuda else 0
        nv.call('segnb_absmax_f32', nv.ptr(self.flat_g), self.total, nv.ptr(out), st)
        return out

    def grads_alias(self):
        """True when every parameter's .grad is already a view of flat_g (accumulate in place)."""
        base = self.flat_g.data_ptr()
        for p in self.param_list():
            g = p.grad
            if g is None:
                return False
            off = self._off[id(p)][0]
            if g.data_ptr() != base + 4 * off:
                return False
        return True

    def begin_backward(self):
        """Decide how this backward meets existing .grad tensors.  Returns True when the kernels should
        accumulate on top of flat_g (every .grad already aliases it: find_optimal_lr never zeroes grads,
        lib/train_utils.py:54-65; zero_grad(set_to_none=False) zeroes them in place); otherwise flat_g is
        cleared first."""
        if getattr(self, 'stepped_in_backward', False):
            import warnings
            warnings.warn('the previous backward updated the parameters in the all-reduce epilogue '
                          '(DataParallel.fuse_optimizer) and no optimizer.step() consumed it: that step was applied anyway')
        self.stepped_in_backward = False
        # data-parallel jobs note whether this backward accumulates on top of earlier gradients (DataParallel sets the
        # flag on the module when it attaches, i.e. before the first backward)
        track = bool(getattr(self, 'track_accumulation', False) or getattr(self.module, '_dp_track_accumulation', False))
        self.accumulation_tracked = track
        if self.grads_alias():
            # accumulating on top of non-zero gradients?  (a data-parallel sync must not reduce them twice)
            self.accumulating = bool(self.grad_absmax() > 0) if track else False
            self.fresh_backward = not self.accumulating
            return True
        self.accumulating = False
        # every .grad None (zero_grad's default): this backward's flat_g IS the step's gradient -- what a data-parallel
        # optimizer fold needs to know (a foreign .grad tensor gets the result added to it afterwards instead)
        self.fresh_backward = all(p.grad is None for p in self.param_list())
        ev, self._prezeroed = getattr(self, '_prezeroed', None), None
        if ev is not None:
            torch.cuda.current_stream(self.flat_g.device).wait_event(ev)      # cleared on the side stream during the forward
        else:
            self.flat_g.zero_()
        return False

    # prezero_grads = False (class attribute): the flat gradient buffer is cleared at the start of backward, on the dependent chain (A/B)
    prezero_grads = True

    def prezero(self, rt, forked=False):
        """(forked: the caller has just made the side stream wait for this one -- no second marker on the main queue)
        Called by a differentiated training forward: when the coming backward will have to clear the flat gradient buffer
        (no .grad aliases it: zero_grad()'s default, torch_train.py:180), clear it NOW on the side stream -- idle during the
        forward -- behind everything issued so far (the optimizer step / logging that read the last gradients).  126 MB for
        ZF_UNET: 18 us off the start of every backward."""
        self._prezeroed = None
        if not self.prezero_grads or self.flat_g is None or self.flat_g.device.type != 'cuda' or self.grads_alias():
            return
        side = rt.side_stream()
        if side is None:
            return
        if not forked:
            nv.call('segnb_stream_fork', rt.stream, side.cuda_stream)
        ev = getattr(self, '_prezero_event', None)
        if ev is None:                     # ONE event, re-recorded every step (destroying an event may wait for it)
            ev = self._prezero_event = torch.cuda.Event()
        with torch.cuda.stream(side):
            self.flat_g.zero_()
            ev.record(side)
        self._prezeroed = ev

    def publish_grads(self, accumulated_in_place):
        """Make parameter.grad reflect flat_g: install views where .grad is None, add into foreign ones."""
        if accumulated_in_place:
            return
        base = self.flat_g.data_ptr()
        views = self._gviews
        for p in self.param_list():
            if not p.requires_grad:
                continue                    # frozen (torch_train_ab.py:245-246): autograd would leave .grad alone too
            view = views[id(p)]
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != base + 4 * self._off[id(p)][0]:
                p.grad.add_(view)
