#!/usr/bin/env python3
"""Timing of the rolling-window kernels with operands recomputed on load against the launches they replace
(32 -> 32 at 224 x 224, bs = 32): tools/tf_bench.py [--reps 30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb import _native as nv
from segnb.engine import ConvOp, Runtime, View


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    N, H, W, C = args.batch, args.size, args.size, 32
    rt = Runtime('cuda', 'bf16')
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    op = ConvOp(rt, w, torch.zeros(C, device='cuda'), [(C, C)], 1, 1, False, True)
    op.pack(H, W)
    mk = lambda: View.alloc(rt, N, H, W, C)
    y0, y0b, a0, y2, g2, dy, dx = mk(), mk(), mk(), mk(), mk(), mk(), mk()
    for v in (y0, y0b, y2, g2):
        v.t.normal_()
    coef = torch.stack([0.5 + torch.rand(C), 0.3 * torch.randn(C), 0.2 * torch.randn(C), 0.5 + torch.rand(C)]).cuda().contiguous()
    bcoef = torch.stack([coef[0].cpu(), 0.05 * torch.randn(C), 0.05 * torch.randn(C)]).cuda().contiguous()
    stats = rt.zeros((16, 2, C), torch.float64)
    sums = rt.zeros((16, 2, C), torch.float64)
    st = torch.cuda.current_stream().cuda_stream
    act = nv.ACT_RELU
    R = args.reps
    t_act = timeit(lambda: nv.call('segnb_bn_act_fwd', rt.code, y0.ptr, y0.ld, N, H, W, C, nv.ptr(coef), act, 0.0, None, a0.ptr,
                                   a0.ld, None, 0, None, 0, None, 0, st), R)
    t_f = timeit(lambda: op.fprop(a0, y2, stats), R)
    t_ftf = timeit(lambda: op.fprop_tf(y0, ConvOp.tf_act(coef, C, act, 0.0), y2, stats), R)
    print('forward : activation pass %.1f + conv %.1f = %.1f us   |  conv with activation on load %.1f us' % (t_act, t_f, t_act + t_f, t_ftf))
    t_ap = timeit(lambda: nv.call('segnb_bn_bwd_apply_direct', rt.code, y2.ptr, y2.ld, N, H, W, C, nv.ptr(coef), nv.ptr(bcoef), act,
                                  0.0, g2.ptr, g2.ld, dy.ptr, dy.ld, None, C, st), R)
    t_d = timeit(lambda: op.dgrad(dy, dx, bn_reduce=(y0b, coef, sums, act, 0.0)), R)
    tfd = ConvOp.tf_bnbwd(y2, coef, bcoef, act, 0.0)
    t_dtf = timeit(lambda: op.dgrad_tf(g2, tfd, dx, bn_reduce=(y0b, coef, sums, act, 0.0)), R)
    print('dgrad   : apply pass %.1f + dgrad(+reduce) %.1f = %.1f us   |  dgrad with apply on load %.1f us' % (t_ap, t_d, t_ap + t_d, t_dtf))
    gw = torch.zeros_like(w)
    t_w = timeit(lambda: op.wgrad(a0, dy, gw, unpack=False), R)
    t_wtf = timeit(lambda: op.wgrad_tf(y0, ConvOp.tf_act(coef, C, act, 0.0), g2, tfd), R)
    t_wtfd = timeit(lambda: op.wgrad_tf(a0, None, g2, tfd), R)
    t_wtfx = timeit(lambda: op.wgrad_tf(y0, ConvOp.tf_act(coef, C, act, 0.0), dy, None), R)
    t_w0 = timeit(lambda: op.wgrad_tf(a0, None, dy, None), R)
    print('wgrad   : plain %.1f us | rolling: plain %.1f, x on load %.1f, dy on load %.1f, both %.1f us' % (t_w, t_w0, t_wtfx, t_wtfd, t_wtf))


if __name__ == '__main__':
    main()
