"""Autograd functions over the fused per-pixel loss kernels (segnb_seg_loss_*).

One streaming pass over (logits fp32 NCHW, target int64) produces every global sum the reference's
binary losses and metrics need (lib/losses.py:7-101, lib/metrics.py:9-43); a 1-thread finalize kernel
turns them into the scalar loss / IoU / accuracy on the device (no host sync); a second pass writes
d(loss)/d(logits) scaled by the upstream gradient (the ``B *`` of torch_train.py:187-188).
"""
import torch

from . import _native as nv

SPEC_FIELDS = ('w_bce', 'w_focal', 'w_jaccard', 'w_sjaccard', 'w_dice', 'smooth', 'eps', 'norm', 'focal_mean')


def make_spec(w_bce=0.0, w_focal=0.0, w_jaccard=0.0, w_sjaccard=0.0, w_dice=0.0, smooth=100.0, eps=1e-7,
              norm=1.0, focal_mean=0):
    return (float(w_bce), float(w_focal), float(w_jaccard), float(w_sjaccard), float(w_dice), float(smooth),
            float(eps), float(norm), int(focal_mean))


def _cspec(spec):
    s = nv.LossSpec()
    for k, v in zip(SPEC_FIELDS, spec):
        setattr(s, k, v)
    return s


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


# optional hook: a data-parallel job all-reduces the 8 global sums here (segnb.dist installs it)
sums_allreduce_hook = None
# data-parallel: the backward seed is B_local; the global-batch loss needs B_total = B_local * world
grad_scale = 1.0


def _prep(logits, target):
    if logits.shape != target.shape:
        raise ValueError('logits %s and target %s must have the same shape'
                         % (tuple(logits.shape), tuple(target.shape)))
    x = logits.detach().contiguous().float()
    t = target.detach()
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return x, t.contiguous()


def reduce_finalize(x, t, spec):
    """-> fin fp32[8] = (loss, soft IoU, pixel accuracy, GI, GU, bce mean, n, -) on the device."""
    sums = torch.zeros(8, dtype=torch.float64, device=x.device)
    fin = torch.empty(8, dtype=torch.float32, device=x.device)
    st = _stream(x)
    nv.call('segnb_seg_loss_reduce', nv.ptr(x), nv.ptr(t), x.numel(), nv.ptr(sums), st)
    if sums_allreduce_hook is not None:
        sums_allreduce_hook(sums)
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
        return fin[0].clone()

    @staticmethod
    def backward(ctx, gout):
        x, t, sums, fin = ctx.saved_tensors
        g = gout.detach().contiguous().float()
        if grad_scale != 1.0:
            g = g * grad_scale
        dx = torch.empty_like(x)
        cs = _cspec(ctx.spec)
        nv.call('segnb_seg_loss_bwd', nv.ptr(x), nv.ptr(t), x.numel(), nv.ptr(sums), nv.ptr(fin), cs, nv.ptr(g),
                nv.ptr(dx), _stream(x))
        return dx, None, None


def seg_loss(logits, target, spec):
    return SegLossFn.apply(logits, target, spec)


def seg_metrics(logits, target):
    """(soft IoU, pixel accuracy) as 0-dim device tensors, one pass."""
    x, t = _prep(logits, target)
    _, fin = reduce_finalize(x, t, make_spec(w_bce=1.0))
    return fin[1].clone(), fin[2].clone()
