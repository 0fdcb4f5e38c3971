#!/usr/bin/env python3
"""A dense layer's data gradient (tiramisu.py:9-20, 16 -> prefix channels) + the BatchNorm backward of the prefix: the stored form
(segnb_conv_fprop_bnreduce + segnb_bn_bwd_apply_fused_direct_acc) against the two launches that never store the gradient
(segnb_conv_fprop_bnsums + segnb_conv_fprop_bnapply), stand-alone, at FCDenseNet103's sizes (256 x 256, bs 8)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb import _native as nv
from segnb.engine import ConvOp, Runtime, View


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    rt = Runtime('cuda', 'bf16')
    for N, H, W, C1 in ((8, 256, 256, 112), (8, 128, 128, 272), (8, 64, 64, 464), (8, 32, 32, 656), (8, 16, 16, 896), (8, 8, 8, 1072),
                        (8, 256, 256, 256)):
        w2 = (torch.randn(16, C1, 3, 3) * 0.05).cuda()
        op = ConvOp(rt, w2, None, [(C1, C1)], 1, 1, False, True)
        op.pack(H, W)
        dyv, y1, g, dx = View.alloc(rt, N, H, W, op.Cop), View.alloc(rt, N, H, W, C1), View.alloc(rt, N, H, W, C1), View.alloc(rt, N, H, W, C1)
        dyv.t.normal_(); y1.t.normal_()
        coef = torch.rand(4, C1, device='cuda') + 0.5
        gamma = torch.rand(C1, device='cuda') + 0.5
        sums = rt.zeros((16, 2, C1), torch.float64)
        bc, dg, db = rt.zeros((3, C1), torch.float32), torch.zeros(C1, device='cuda'), torch.zeros(C1, device='cuda')
        st = rt.stream
        t_plain = timed(lambda: op.dgrad(dyv, g))
        t_red = timed(lambda: op.dgrad(dyv, g, bn_reduce=(y1, coef, sums, nv.ACT_RELU, 0.0)))
        t_app = timed(lambda: nv.call('segnb_bn_bwd_apply_fused_direct_acc', rt.code, y1.ptr, y1.ld, N, H, W, C1, C1, nv.ptr(coef),
                                      nv.ptr(sums), nv.ptr(gamma), nv.ptr(bc), nv.ptr(dg), nv.ptr(db), 1, None, nv.ACT_RELU, 0.0,
                                      g.ptr, g.ld, dx.ptr, dx.ld, st))
        t_sums = timed(lambda: op.dgrad_bnsums(dyv, H, W, (y1, coef, sums, nv.ACT_RELU, 0.0)))
        ep = nv.BnApplyEpilogue(y1.ptr, y1.ld, nv.ptr(coef), nv.ptr(sums), nv.ptr(gamma), C1, float(N * H * W), nv.ptr(bc), nv.ptr(dg),
                                nv.ptr(db), nv.ACT_RELU, 0.0, dx.ptr, dx.ld, 1)
        t_ap2 = timed(lambda: op.dgrad_bnapply(dyv, H, W, ep))
        mb = N * H * W * C1 * 2 / 1e6
        print('%dx%dx%dx%-4d tensor %6.1f MB: plain dgrad %6.1f | dgrad+reduce %6.1f + apply_acc %6.1f = %6.1f us | sums %6.1f + apply %6.1f = %6.1f us'
              % (N, H, W, C1, mb, t_plain, t_red, t_app, t_red + t_app, t_sums, t_ap2, t_sums + t_ap2))


if __name__ == '__main__':
    main()
