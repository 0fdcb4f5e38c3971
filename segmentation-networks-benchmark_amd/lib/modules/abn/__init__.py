"""In-Place Activated BatchNorm on the MI355X kernels -- drop-in for the reference's ``lib.modules.abn``
(/root/reference/lib/modules/abn/bn.py:23-103, functions.py:62-122).

The reference delegates to the un-vendored ``inplace_abn`` CUDA extension (functions.py:1: mean_var / forward /
edz_eydz / backward).  Here the same four phases are segnb_bn_stats + segnb_bn_finalize / segnb_bn_act_fwd /
segnb_bn_act_bwd_reduce / segnb_bn_bwd_apply.  Inside LinkNet34 the module is only a parameter holder (the
executor fuses BN + LeakyReLU behind each convolution); called on its own it runs the same kernels, with the NCHW <->
NHWC moves as launches of the library too, and writes its result into the input's storage like the reference.

Affine form.  The un-vendored backend is absent, so which scale it applies cannot be
read off the reference.  The API generation functions.py binds -- ``forward(x, mean, var, weight, bias, affine, eps)``,
``edz_eydz(z, dz, weight, bias, affine, eps)``, ``backward(z, dz, var, weight, bias, edz, eydz, affine, eps)`` -- is
mapillary's first release, whose kernels scale by ``abs(weight) + eps`` (invertibility of the in-place form) and
return ``dweight = sign(weight) * sum(dz * yhat)``.  Both forms are offered:
  * ``affine_form='gamma'`` (default): y = yhat * weight + bias -- what torch.nn.BatchNorm2d computes, what the
    LinkNet34 executor fuses, and what every golden of this repository was produced with (the stand-in of
    tests/golden/make_golden.py is BatchNorm2d + LeakyReLU);
  * ``affine_form='abs_eps'`` (or SEGNB_ABN_AFFINE=abs_eps): y = yhat * (|weight| + eps) + bias with the backend's
    gradient.  At the weight = 1 initialisation the two differ by the factor 1 + eps (1e-5 relative per layer), for
    negative or near-zero weights outright.  Inside LinkNet34's fused plan the same form is served by two C-element
    launches around the BatchNorm entry points (segnb_abn_scale / segnb_abn_dscale, segnb.net.conv_unit).
``InPlaceABNSync`` is instantiated by no model in the reference and is not provided.
"""
import os

from collections import OrderedDict

import torch
import torch.nn as nn

from segnb import _native as nv
from segnb import convplan as cp

ACT_LEAKY_RELU, ACT_ELU, ACT_NONE = 'leaky_relu', 'elu', 'none'


class ABN(nn.Sequential):
    """BatchNorm2d + activation as two torch modules (bn.py:23-44)."""

    def __init__(self, num_features, activation=nn.ReLU(inplace=True), **kwargs):
        super(ABN, self).__init__(OrderedDict([('bn', nn.BatchNorm2d(num_features, **kwargs)), ('act', activation)]))


class _ABNFn(torch.autograd.Function):
    """The four phases of functions.py:62-122 on the HIP kernels.  No torch operator touches the activations: the NCHW
    tensors of the module interface are moved to / from the kernels' NHWC layout by segnb_pack_input_nchw /
    segnb_nhwc_to_nchw_f32, and -- as in the reference (functions.py:92 ``ctx.mark_dirty(x)``) -- the result is written
    INTO x's storage when autograd allows it (x is not a leaf that requires grad), so the module allocates no output."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, slope, abs_form=False):
        N, C, H, W = x.shape
        w_param = weight
        if abs_form and weight is not None:
            # the backend's effective scale (a C-element parameter op; the activations stay on the kernels)
            weight = weight.detach().abs() + eps
        Cp = cp.pad8(C)
        dev, st = x.device, (torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0)
        y = torch.empty((N, H, W, Cp), dtype=torch.float32, device=dev)
        nv.call('segnb_pack_input_nchw', nv.ptr(x), N, C, H, W, nv.ptr(y), nv.F32, Cp, Cp, st)
        stats = torch.zeros((16, 2, Cp), dtype=torch.float64, device=dev)
        coef = torch.zeros((4, Cp), dtype=torch.float32, device=dev)
        if training:
            nv.call('segnb_bn_stats', nv.F32, nv.ptr(y), Cp, N, H, W, Cp, nv.ptr(stats), st)
        nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * H * W),
                nv.ptr(weight.detach() if weight is not None else None),
                nv.ptr(bias.detach() if bias is not None else None), eps, momentum, nv.ptr(running_mean),
                nv.ptr(running_var), None,
                1 if training else 0, nv.ptr(coef), st)
        out = torch.empty_like(y)
        nv.call('segnb_bn_act_fwd', nv.F32, nv.ptr(y), Cp, N, H, W, Cp, nv.ptr(coef), act, slope, None, nv.ptr(out), Cp,
                None, 0, None, 0, None, 0, st)
        ctx.training = bool(training)
        ctx.save_for_backward(y, coef, weight if weight is not None else coef.new_empty(0),
                              w_param if (abs_form and w_param is not None) else coef.new_empty(0))
        ctx.cfg = (N, C, H, W, Cp, act, slope, weight is not None)
        ctx.abs_form = bool(abs_form and w_param is not None)
        in_place = not (x.is_leaf and x.requires_grad)
        res = x if in_place else torch.empty_like(x)
        nv.call('segnb_nhwc_to_nchw_f32', nv.F32, nv.ptr(out), Cp, N, H, W, C, nv.ptr(res), st)
        if in_place:
            ctx.mark_dirty(x)
        return res

    @staticmethod
    def backward(ctx, gout):
        y, coef, weight, w_param = ctx.saved_tensors
        N, C, H, W, Cp, act, slope, affine = ctx.cfg
        dev, st = y.device, (torch.cuda.current_stream(y.device).cuda_stream if y.is_cuda else 0)
        gout = gout.detach().contiguous().float()
        g = torch.empty((N, H, W, Cp), dtype=torch.float32, device=dev)
        nv.call('segnb_pack_input_nchw', nv.ptr(gout), N, C, H, W, nv.ptr(g), nv.F32, Cp, Cp, st)
        dz = torch.empty_like(g)
        sums = torch.zeros((16, 2, Cp), dtype=torch.float64, device=dev)
        bcoef = torch.zeros((3, Cp), dtype=torch.float32, device=dev)
        dgamma = torch.zeros(C, dtype=torch.float32, device=dev)
        dbeta = torch.zeros(C, dtype=torch.float32, device=dev)
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dev)
        nv.call('segnb_bn_act_bwd_reduce', nv.F32, nv.ptr(y), Cp, N, H, W, Cp, nv.ptr(coef), act, slope, None, nv.ptr(g),
                Cp, None, 0, None, 0, nv.ptr(dz), Cp, nv.ptr(sums), None, 0, st)
        if not ctx.training:
            # inference-mode backward of the reference (functions.py:113-116: edz = eydz = 0): a plain affine map,
            # dx = dz * gamma * rstd; dgamma / dbeta are still the sums (the training formula is wrong here).
            # As launches: sums -> dgamma / dbeta, then dz scaled per channel by coef row 0 = gamma * rstd of the running
            # statistics (segnb_bn_act_fwd with a coefficient table whose shift row is zero, no activation)
            nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, float(N * H * W),
                    nv.ptr(weight.detach() if affine else None), nv.ptr(coef), nv.ptr(bcoef), nv.ptr(dgamma),
                    nv.ptr(dbeta), 0, st)
            scale_only = torch.zeros_like(coef)
            scale_only[0].copy_(coef[0])
            nv.call('segnb_bn_act_fwd', nv.F32, nv.ptr(dz), Cp, N, H, W, Cp, nv.ptr(scale_only), nv.ACT_NONE, 0.0, None,
                    nv.ptr(dz), Cp, None, 0, None, 0, None, 0, st)
        else:
            nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, float(N * H * W),
                    nv.ptr(weight.detach() if affine else None), nv.ptr(coef), nv.ptr(bcoef), nv.ptr(dgamma),
                    nv.ptr(dbeta), 0, st)
            nv.call('segnb_bn_bwd_apply', nv.F32, nv.ptr(y), Cp, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), nv.ptr(dz), Cp,
                    nv.ptr(dz), Cp, None, C, st)
        nv.call('segnb_nhwc_to_nchw_f32', nv.F32, nv.ptr(dz), Cp, N, H, W, C, nv.ptr(dx), st)
        if ctx.abs_form:
            # d(|w| + eps)/dw as the backend signs it: +1 for w > 0, -1 otherwise
            dgamma = torch.where(w_param.detach() > 0, dgamma, -dgamma)
        return (dx, dgamma if affine else None, dbeta if affine else None, None, None, None, None, None, None, None, None)


class InPlaceABN(nn.Module):
    """InPlace Activated Batch Normalization (bn.py:47-103): parameters weight/bias, buffers running_mean/var."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation='leaky_relu', slope=0.01,
                 affine_form=None):
        super(InPlaceABN, self).__init__()
        if activation not in (ACT_LEAKY_RELU, ACT_ELU, ACT_NONE):
            raise ValueError('activation must be one of leaky_relu, elu, none (bn.py:59), got %r' % (activation,))
        if affine_form is None:
            affine_form = os.environ.get('SEGNB_ABN_AFFINE', 'gamma')
        if affine_form not in ('gamma', 'abs_eps'):
            raise ValueError("affine_form must be 'gamma' or 'abs_eps', got %r" % (affine_form,))
        self.affine_form = affine_form
        self.num_features, self.affine, self.eps, self.momentum = num_features, affine, eps, momentum
        self.activation, self.slope = activation, slope
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))

    def reset_parameters(self):
        nn.init.constant_(self.running_mean, 0)
        nn.init.constant_(self.running_var, 1)
        if self.affine:
            nn.init.constant_(self.weight, 1)
            nn.init.constant_(self.bias, 0)

    def forward(self, x):
        act = nv.ACT_LEAKY if self.activation == ACT_LEAKY_RELU else nv.ACT_NONE
        y = _ABNFn.apply(x.contiguous().float(), self.weight, self.bias, self.running_mean, self.running_var,
                         self.training, self.momentum, self.eps, act, self.slope, self.affine_form == 'abs_eps')
        # 'elu' is used by no model of the reference: BatchNorm on the HIP kernels, the ELU as a torch op on top
        return torch.nn.functional.elu(y) if self.activation == ACT_ELU else y

    def __repr__(self):
        return '%s(%d, eps=%g, momentum=%g, affine=%s, activation=%s slope=%g)' % (
            self.__class__.__name__, self.num_features, self.eps, self.momentum, self.affine, self.activation,
            self.slope)
