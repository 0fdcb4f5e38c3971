"""Autograd functions over the fused per-pixel loss kernels (segnb_seg_loss_*).

One streaming pass over (logits fp32 NCHW, target int64) produces every global sum the reference's
binary losses and metrics need (lib/losses.py:7-101, lib/metrics.py:9-43); a 1-thread finalize kernel
turns them into the scalar loss / IoU / accuracy on the device (no host sync); a second pass writes
d(loss)/d(logits) scaled by the upstream gradient (the ``B *`` of torch_train.py:187-188).

The metrics of the training loop are called on the SAME (outputs, y) right after the loss
(torch_train.py:185,209-210): ``seg_metrics`` answers from the sums the loss launch already produced -- zero
extra passes over the logits.
"""
import weakref

import os

import torch

from . import _native as nv

SPEC_FIELDS = ('w_bce', 'w_focal', 'w_jaccard', 'w_sjaccard', 'w_dice', 'smooth', 'eps', 'norm', 'focal_mean',
               'bce_sum', 'focal_gamma')


def make_spec(w_bce=0.0, w_focal=0.0, w_jaccard=0.0, w_sjaccard=0.0, w_dice=0.0, smooth=100.0, eps=1e-7,
              norm=1.0, focal_mean=0, bce_sum=0, focal_gamma=2.0):
    return (float(w_bce), float(w_focal), float(w_jaccard), float(w_sjaccard), float(w_dice), float(smooth),
            float(eps), float(norm), int(focal_mean), int(bce_sum), float(focal_gamma))


def _cspec(spec):
    s = nv.LossSpec()
    for k, v in zip(SPEC_FIELDS, spec):
        setattr(s, k, v)
    return s


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


class DataParallelHooks(object):
    """What a data-parallel job plugs into the loss path (segnb.dist.DataParallel installs / removes them):
    sums_allreduce -- all-reduces the 8 global sums; grad_scale -- the backward seed is B_local, the global-batch
    loss needs B_total = B_local * world."""
    sums_allreduce = None
    grad_scale = 1.0

    @classmethod
    def reset(cls):
        cls.sums_allreduce, cls.grad_scale = None, 1.0


def _prep(logits, target):
    if logits.shape != target.shape:
        raise ValueError('logits %s and target %s must have the same shape'
                         % (tuple(logits.shape), tuple(target.shape)))
    x = logits.detach().contiguous().float()
    t = target.detach()
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return x, t.contiguous()


# (weakref to the logits tensor, its version, weakref to the target, its version, fin) of the last reduction
_last = None


def _remember(logits, target, fin):
    global _last
    try:
        _last = (weakref.ref(logits), logits._version, weakref.ref(target), target._version, fin)
    except TypeError:
        _last = None


def _recall(logits, target):
    if _last is None:
        return None
    rl, vl, rt, vt, fin = _last
    if rl() is logits and rt() is target and logits._version == vl and target._version == vt:
        return fin
    return None


# A model may name the buffer it wants d(loss)/d(logits) written into (its recorded backward list reads the gradient from ONE
# persistent tensor): {data_ptr of the logits it returned: buffer}.  The loss backward then writes there instead of into a fresh
# tensor that the model would have to copy (6 MB and a launch per step at 224 x 224 bs=32).  One entry: the latest forward.
_grad_buffers = {}


def register_grad_buffer(logits, buf):
    _grad_buffers.clear()
    _grad_buffers[logits.data_ptr()] = buf


def _grad_buffer_for(x):
    """The registered buffer, handed out AT MOST ONCE per forward: a second loss term on the same logits
    (``BCEWithSigmoidLoss()(out, y) + DiceLoss()(out, y)``), a second seg_loss call or a retained graph gets a fresh tensor --
    two backward nodes returning the same storage would overwrite each other and autograd would add the tensor to itself
    (ADVICE r4).  The model copies when the gradient it receives is not its buffer (ZF_UNET.backward)."""
    buf = _grad_buffers.pop(x.data_ptr(), None)
    if buf is not None and buf.shape == x.shape and buf.dtype == x.dtype and buf.device == x.device:
        return buf
    return None


# _ONE_LAUNCH = False (module attribute): zero fill, segnb_seg_loss_reduce and segnb_seg_loss_finalize as three launches (A/B)
_ONE_LAUNCH = True
_loss_work = {}       # (device index, stream) -> 128 zeroed doubles: the sums + ticket of segnb_seg_loss_reduce_finalize


def reduce_finalize(x, t, spec):
    """-> fin fp32[8] = (loss, soft IoU, pixel accuracy, GI, GU, bce mean, n, -) on the device."""
    fin = torch.empty(8, dtype=torch.float32, device=x.device)
    st = _stream(x)
    if DataParallelHooks.sums_allreduce is None and _ONE_LAUNCH:
        # one device: zero fill + reduction + finalize are ONE launch (the work buffer is left zeroed by the launch itself; one
        # per stream: two losses evaluated on different streams must not share sums)
        key = (x.device.index, st)
        work = _loss_work.get(key)
        if work is None:
            work = _loss_work[key] = torch.zeros(128, dtype=torch.float64, device=x.device)
        try:
            nv.call('segnb_seg_loss_reduce_finalize', nv.ptr(x), nv.ptr(t), x.numel(), _cspec(spec), nv.ptr(work), nv.ptr(fin), st)
        except BaseException:
            # the launch itself re-zeroes the buffer; one that failed may not have: the next loss on this stream starts from
            # a fresh, zeroed one instead of inheriting partial sums or a ticket
            _loss_work.pop(key, None)
            raise
        return work, fin
    sums = torch.zeros(8, dtype=torch.float64, device=x.device)
    nv.call('segnb_seg_loss_reduce', nv.ptr(x), nv.ptr(t), x.numel(), float(spec[10]), nv.ptr(sums), st)
    if DataParallelHooks.sums_allreduce is not None:
        DataParallelHooks.sums_allreduce(sums)
    cs = _cspec(spec)
    nv.call('segnb_seg_loss_finalize', nv.ptr(sums), cs, nv.ptr(fin), st)
    return sums, fin


class SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, spec):
        x, t = _prep(logits, target)
        sums, fin = reduce_finalize(x, t, spec)
        ctx.spec = spec
        ctx.save_for_backward(x, t, sums, fin)
        SegLossFn.last_fin = fin
        return fin[0]                 # (a view of this call's own result vector: no copy launch)

    @staticmethod
    def backward(ctx, gout):
        x, t, sums, fin = ctx.saved_tensors
        g = gout.detach().contiguous().float()
        if DataParallelHooks.grad_scale != 1.0:
            g = g * DataParallelHooks.grad_scale
        dx = _grad_buffer_for(x)
        if dx is None:
            dx = torch.empty_like(x)
        cs = _cspec(ctx.spec)
        nv.call('segnb_seg_loss_bwd', nv.ptr(x), nv.ptr(t), x.numel(), nv.ptr(sums), nv.ptr(fin), cs, nv.ptr(g),
                nv.ptr(dx), _stream(x))
        return dx, None, None


def seg_loss(logits, target, spec):
    out = SegLossFn.apply(logits, target, spec)
    _remember(logits, target, SegLossFn.last_fin)
    SegLossFn.last_fin = None
    return out


def seg_metrics(logits, target):
    """(soft IoU, pixel accuracy) as 0-dim device tensors.  When the loss was just computed on these very tensors the
    values come from its sums (no launch); otherwise one pass."""
    fin = _recall(logits, target)
    if fin is None:
        x, t = _prep(logits, target)
        _, fin = reduce_finalize(x, t, make_spec(w_bce=1.0))
        _remember(logits, target, fin)
    return fin[1].clone(), fin[2].clone()


class SegLossMapFn(torch.autograd.Function):
    """reduce=False: the per-pixel loss map (kind 0 = double-sigmoid BCE element, 1 = focal element)."""

    @staticmethod
    def forward(ctx, logits, target, kind, gamma):
        x, t = _prep(logits, target)
        out = torch.empty_like(x)
        nv.call('segnb_seg_loss_map', nv.ptr(x), nv.ptr(t), x.numel(), kind, float(gamma), nv.ptr(out), _stream(x))
        ctx.cfg = (kind, float(gamma))
        ctx.save_for_backward(x, t)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, t = ctx.saved_tensors
        kind, gamma = ctx.cfg
        g = gout.detach().contiguous().float()
        dx = torch.empty_like(x)
        nv.call('segnb_seg_loss_map_bwd', nv.ptr(x), nv.ptr(t), x.numel(), kind, gamma, nv.ptr(g), nv.ptr(dx), _stream(x))
        return dx, None, None, None


def seg_loss_map(logits, target, kind, gamma=2.0):
    return SegLossMapFn.apply(logits, target, kind, gamma)
