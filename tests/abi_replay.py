"""Teacher-forced plan parity (test infrastructure).

A random-init segmentation net is a chaotic map: one ReLU whose pre-activation sits within an fp32 ulp of zero flips
between two correct fp32 implementations and moves every upstream gradient by 1e-3..1e-2 (measured on MI355X:
FCDenseNet fixture, pixel 1443 / channel 18 of one BatchNorm, z = -5.2e-7 vs +3.2e-7).  End-to-end gradient
comparisons therefore cannot be tight.  This harness is: run one training step of a model on the ABI emulator
(oracle/abi_emulator.py) and record, for EVERY C-ABI call of the plan, every floating tensor argument before and
after the call; then run the same step through libsegnb_hip.so, overwrite the tensor arguments of each call with the
emulator's pre-call values (teacher forcing) and compare what the HIP kernel produced with what the emulator
produced.  No error accumulates from call to call, so every launch of the real plan (real shapes, slices, strides,
geometries) is held to the per-op tolerance.
"""
import bisect

import torch

from oracle import abi_emulator
from segnb import _native as nv
import segnb.engine as E


BIG_BYTES = 4 << 20
BIG_WRITERS = ('segnb_unpack_wgrad', 'segnb_bn_bwd_finalize', 'segnb_head_bwd', 'segnb_head_bn_bwd', 'segnb_sgd_step')


class _Recorder(object):
    def __init__(self):
        self.registry = {}          # storage base pointer -> (nbytes, tensor)
        self.calls = []
        self.forced = None          # emulator record to force / compare against (GPU pass)
        self.report = []

    def reg(self, t):
        if t is not None and torch.is_tensor(t):
            st = t.untyped_storage()
            self.registry[st.data_ptr()] = (st.nbytes(), t)

    def _tensor_args(self, args):
        starts = sorted(self.registry)
        out = []
        for pos, a in enumerate(args):
            if isinstance(a, int) and not isinstance(a, bool) and a > (1 << 24):
                i = bisect.bisect_right(starts, a) - 1
                if i >= 0 and a < starts[i] + self.registry[starts[i]][0]:
                    t = self.registry[starts[i]][1]
                    if t.is_floating_point():
                        full = torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage())
                        out.append((pos, full))
        return out

    @staticmethod
    def checksum(t):
        return torch.stack([t.sum(dtype=torch.float64), t.abs().sum(dtype=torch.float64)]).cpu()

    @staticmethod
    def snap(t, name, post):
        """Small storages are kept whole; big ones (the flat parameter / gradient buffers of a 20M-parameter net,
        touched by hundreds of calls) as (sum, sum|.|) checksums after the calls that write into them."""
        if t.numel() * t.element_size() > BIG_BYTES:
            if post and name in BIG_WRITERS:
                return ('sum', _Recorder.checksum(t))
            return ('skip', None)
        return ('full', t.detach().clone().cpu())


def _cmp(name, pos, ref, got, tol, max_outliers, atol=0.0, nslab=(1, 1)):
    ref, got = ref.double(), got.double().cpu()
    if (name == 'segnb_conv_wgrad' and pos == 4) or (name == 'segnb_conv_wgrad_tf' and pos == 6) or \
            (name == 'segnb_conv_wgrad_bnapply' and pos == 12):
        # the result is slab 0; the other slabs are scratch (and their COUNT differs between emulator and device)
        ref, got = ref.view(nslab[0], -1)[0], got.view(nslab[1], -1)[0]
    if ref.dtype == torch.float64 and ref.numel() % (2 * abi_emulator.REPL) == 0 and name in _REPLICATED.get(pos, ()):
        ref, got = ref.view(abi_emulator.REPL, -1).sum(0), got.view(abi_emulator.REPL, -1).sum(0)
    scale = float(ref.abs().max())
    if scale == 0.0:
        return float(got.abs().max()) == 0.0, 0, '0'
    err = (ref - got).abs()
    badmask = err > tol * scale + atol
    bad = int(badmask.sum())
    detail = ''
    if bad > max_outliers:
        idx = badmask.reshape(-1).nonzero().reshape(-1)[:6].tolist()
        detail = ' first bad (index: emulator, product): ' + ', '.join(
            '%d: %.6e, %.6e' % (i, float(ref.reshape(-1)[i]), float(got.reshape(-1)[i])) for i in idx)
    return bad <= max_outliers, bad, '%.3e%s' % (float(err.max()) / scale, detail)


# (argument position -> entry points) whose fp64 argument is a [REPL][2][Cp] replicated accumulator: the emulator
# fills replica 0, the HIP kernels spread blocks over all 16; only the sum over replicas is defined by the ABI
_REPLICATED = {7: ('segnb_conv_fprop', 'segnb_bn_stats', 'segnb_bn_stats_ld'), 19: ('segnb_bn_act_bwd_reduce',),
               17: ('segnb_head_bn_bwd', 'segnb_bn_act_bwd_reduce_add'), 13: ('segnb_bn_act_fwd_stats',), 20: ('segnb_head_conv_bwd',),
               8: ('segnb_conv_fprop_tf',), 0: ()}


def run_step(model, x, y, loss_fn, device, dtype):
    """One torch_train.py:180-190 step body (without the optimizer); returns {parameter name: gradient}."""
    model.set_compute_dtype(dtype)
    model.to(device).train()
    out = model(x.to(device))
    loss = loss_fn(out, y.to(device))
    (x.shape[0] * loss).backward()
    if device != 'cpu':
        torch.cuda.synchronize()
    return {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}


def replay(make_model, x, y, loss_fn, dtype, device='cuda',
           flip_ops=('segnb_bn_act_fwd', 'segnb_bn_act_bwd_reduce', 'segnb_bn_act_bwd_reduce_add', 'segnb_bn_bwd_apply', 'segnb_bn_bwd_apply_direct',
                     'segnb_bn_bwd_finalize',
                     'segnb_bn_fwd_fused', 'segnb_bn_bwd_apply_fused', 'segnb_bn_bwd_apply_fused_direct',
                     'segnb_bn_fwd_fused_head', 'segnb_head_bn_bwd', 'segnb_bn_fwd_fused_ld', 'segnb_bn_act_fwd_stats')):
    """Returns (number of calls, list of failure strings).  device='cpu' replays the emulator against itself (a
    self-test of this harness that runs without a GPU)."""
    rec = _Recorder()
    orig_ptr, orig_call, orig_vptr = nv.ptr, nv.call, E.View.ptr
    # the decoder blocks' data gradient by segment on BOTH sides (the device would segment only the sizes its fast kernel
    # serves, the emulator every size: the two call sequences must be the same)
    orig_force = E.UpCatConvOp.force_segmented
    E.UpCatConvOp.force_segmented = True
    # ... and the upsampled copy materialised on both sides (the device reads the low-resolution tensor only where
    # segnb_conv_upcat_ok finds kernels for the channel counts, the emulator everywhere; the virtual concat has its own per-op
    # test against the materialised form, test_hip_ops.py::test_upcat_segmented_backward_vs_torch)
    orig_vcat = E.UpCatConvOp.virtual_concat
    E.UpCatConvOp.virtual_concat = False

    def ptr(t, offset_elems=0):
        rec.reg(t)
        return orig_ptr(t, offset_elems)

    def call(name, *args):
        targs = rec._tensor_args(args)
        if rec.forced is None:
            pre = [(p, rec.snap(t, name, False)) for p, t in targs]
            orig_call(name, *args)
            post = [(p, rec.snap(t, name, True)) for p, t in targs]
            rec.calls.append((name, pre, post, [a if isinstance(a, int) else None for a in args]))
            return
        idx = len(rec.calls)
        # a weight gradient behind segnb_wgrad_target_arm delivers into the parameter's gradient (inside the target struct, not an
        # argument): its workspace argument is scratch afterwards on both sides -- not compared; the published gradients are, at
        # the end of the step
        armed, rec.armed = getattr(rec, 'armed', False), name == 'segnb_wgrad_target_arm'
        rname, rpre, rpost, _ = rec.forced[idx]
        assert rname == name, 'call %d: emulator ran %s, product runs %s' % (idx, rname, name)
        for (p, t), (rp, (kind, rt)) in zip(targs, rpre):
            assert p == rp, (idx, name, p, rp)
            if kind == 'full':
                if t.numel() != rt.numel():
                    # the weight-gradient workspace: slab counts differ between emulator and device; its
                    # pre-state is (consumed) zeros or dead partials on both sides
                    assert (name in ('segnb_conv_wgrad', 'segnb_unpack_wgrad') and p in (0, 4)) or \
                        (name == 'segnb_conv_wgrad_tf' and p == 6) or (name == 'segnb_conv_wgrad_bnapply' and p == 12), (idx, name, p)
                    continue
                t.copy_(rt.to(t.device))
        orig_call(name, *args)
        tol = 2e-2 if dtype == 'bf16' else 2e-4
        for (p, t), (rp, (kind, rt)) in zip(targs, rpost):
            loose = name in flip_ops
            if kind == 'skip' or (name == 'segnb_unpack_wgrad' and p == 0):
                continue            # (the workspace after an unpack: zeros or dead partials, slab counts differ)
            if armed and ((name in ('segnb_conv_wgrad',) and p == 4) or (name == 'segnb_conv_wgrad_tf' and p == 6) or
                          (name == 'segnb_conv_wgrad_bnapply' and p == 12) or (name == 'segnb_conv_wgrad_upcat' and p == 5)):
                continue            # (scratch: see above)
            if kind == 'sum':
                got = rec.checksum(t)
                if float((got - rt).abs().max()) > 5 * tol * float(rt[1]) + 1e-30:
                    rec.report.append('#%d %s arg %d: checksum (sum, abs) %s vs emulator %s' % (
                        idx, name, p, got.tolist(), rt.tolist()))
                continue
            atol = 0.0
            if name == 'segnb_bn_act_bwd_reduce' and p == 19:
                # channel sums that are analytically zero hold pure cancellation noise on both sides: floor at
                # 1e-6 of sum|dz| (dz = argument 17 of the same call)
                dz = [r for q, (k, r) in rpost if q == 17 and k == 'full']
                atol = 1e-6 * float(dz[0].double().abs().sum()) if dz else 0.0
            if name == 'segnb_bn_act_bwd_reduce_add' and p == 17:       # (the same floor; dz = argument 15)
                dz = [r for q, (k, r) in rpost if q == 15 and k == 'full']
                atol = 1e-6 * float(dz[0].double().abs().sum()) if dz else 0.0
            if name == 'segnb_head_conv_bwd' and p == 20:       # (the same floor; dz = argument 16)
                dz = [r for q, (k, r) in rpost if q == 16 and k == 'full']
                atol = 1e-6 * float(dz[0].double().abs().sum()) if dz else 0.0
            nslab = (rec.forced[idx][3][5], args[5]) if name == 'segnb_conv_wgrad' else (1, 1)
            if name == 'segnb_conv_wgrad_tf':
                nslab = (rec.forced[idx][3][7], args[7])
            if name == 'segnb_conv_wgrad_bnapply':
                nslab = (rec.forced[idx][3][13], args[13])
            ok, bad, worst = _cmp(name, p, rt, t, tol * (5 if loose and rt.dtype == torch.float64 else 1),
                                  3 if loose else 0, atol, nslab)
            if not ok:
                rec.report.append('#%d %s arg %d: %d/%d elements beyond %.0e of max (worst %s)' % (
                    idx, name, p, bad, t.numel(), tol, worst))
        rec.calls.append((name, None, None, None))

    overlap = E.Runtime.overlap_wgrad
    E.Runtime.overlap_wgrad = False          # the harness reads results right after each call: one stream
    from lib.models import zf_unet as _zf
    cplan = _zf._ZFUnetPlan.use_cplan
    _zf._ZFUnetPlan.use_cplan = False        # every call must pass through the patched nv.call (no C-side replay)
    from segnb import net as _net
    cplan_net = _net.HipNet.use_cplan
    _net.HipNet.use_cplan = False
    nv.ptr = ptr
    nv.call = call
    E.View.ptr = property(lambda self: (rec.reg(self.t), orig_vptr.fget(self))[1])
    try:
        nv.set_backend_for_testing(abi_emulator.AbiEmulator())
        g_emu = run_step(make_model(), x, y, loss_fn, 'cpu', dtype)
        rec.forced, rec.calls = rec.calls, []
        rec.registry.clear()
        if device != 'cpu':
            nv.set_backend_for_testing(None)
        g_hip = run_step(make_model(), x, y, loss_fn, device, dtype)
        assert len(rec.calls) == len(rec.forced), (len(rec.calls), len(rec.forced))
        # the gradients the teacher-forced product pass published vs the emulator's (covers the batched unpack
        # tables, whose tensor arguments are hidden inside a device job table)
        gtol = 2e-2 if dtype == 'bf16' else 5e-4
        for n in g_emu:
            scale = float(g_emu[n].abs().max())
            err = float((g_emu[n] - g_hip[n]).abs().max())
            if err > gtol * scale + 1e-9:
                rec.report.append('final gradient %s: max err %.3e of max %.3e' % (n, err, scale))
    finally:
        nv.set_backend_for_testing(None)
        nv.ptr, nv.call, E.View.ptr = orig_ptr, orig_call, orig_vptr
        E.UpCatConvOp.force_segmented = orig_force
        E.UpCatConvOp.virtual_concat = orig_vcat
        E.Runtime.overlap_wgrad = overlap
        _zf._ZFUnetPlan.use_cplan = cplan
        _net.HipNet.use_cplan = cplan_net
    return len(rec.calls), rec.report
