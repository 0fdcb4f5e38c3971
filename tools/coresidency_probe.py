#!/usr/bin/env python3
"""Does an HBM-bound pass run faster beside the weight-gradient stream when its blocks FIT beside a weight-gradient block on a
CU?  conv_wgrad_s1x9_kernel's 64 x 64-tile blocks hold 8 waves x 203 registers: 96 registers per SIMD lane stay free.  A
torch device copy (a kernel of ~30 registers) can become resident on those CUs, segnb_bn_bwd_apply (155 registers) cannot.
Times both, alone and beside a stream of wide weight gradients sized for half of the CUs (the training step's configuration).
Round 5 (profiles/r05_ab.txt): apply pass 26.8 us alone / 38.4 beside (x 1.43), copy 15.2 / 18.7 (x 1.23); a <= 96-register form
of the apply pass measured 25.5 / 31.8 (x 1.25) here -- and +0.7 % on the training step, so it was not kept.

    python tools/coresidency_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb import _native as nv
from segnb.engine import ConvOp, Runtime, View


def main():
    rt = Runtime('cuda', 'bf16')
    N, hw, ci, co = 32, 28, 256, 256
    wt = torch.randn(co, ci, 3, 3, device='cuda') * 0.05
    op = ConvOp(rt, wt, torch.zeros(co, device='cuda'), [(ci, ci)], 1, 1, False, True)
    op.pack(hw, hw)
    xv = View.alloc(rt, N, hw, hw, op.Cip)
    xv.t.normal_()
    dyv = View.alloc(rt, N, hw, hw, op.Cop)
    dyv.t.normal_()
    gw = torch.zeros_like(wt)
    side = torch.cuda.Stream()
    # the HBM-bound passes: 112 x 112 x 64 channels at bs = 32 (51 MB per tensor)
    n2, h2, c2 = 32, 112, 64
    y = View.alloc(rt, n2, h2, h2, c2)
    y.t.normal_()
    dz = View.alloc(rt, n2, h2, h2, c2)
    dz.t.normal_()
    dy = View.alloc(rt, n2, h2, h2, c2)
    coef = torch.stack([torch.ones(c2), torch.zeros(c2), torch.zeros(c2), torch.ones(c2)]).cuda().contiguous()
    bcoef = torch.stack([torch.ones(c2), torch.zeros(c2), torch.zeros(c2)]).cuda().contiguous()

    def apply_pass():
        nv.call('segnb_bn_bwd_apply', rt.code, y.ptr, y.ld, n2, h2, h2, c2, nv.ptr(coef), nv.ptr(bcoef), dz.ptr, dz.ld, dy.ptr,
                dy.ld, None, c2, rt.stream)

    def copy_pass():          # the same bytes read (2 tensors) is not what a copy does; compare RATIOS alone / beside
        dy.t.copy_(dz.t)

    def timed(fn, reps, beside):
        torch.cuda.synchronize()
        if beside:
            with torch.cuda.stream(side):
                for _ in range(reps * 3):
                    op.wgrad(xv, dyv, gw, unpack=False)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    old = torch.cuda.current_stream()
    for name, fn in (('bn_bwd_apply (155 VGPRs)', apply_pass), ('torch copy (small kernel)', copy_pass)):
        for _ in range(3):
            fn()
        alone = timed(fn, 40, False)
        beside = timed(fn, 40, True)
        print('%-28s alone %7.1f us   beside the weight-gradient stream %7.1f us   x %.2f' % (name, alone, beside, beside / alone))
    w_alone = None
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        op.wgrad(xv, dyv, gw, unpack=False)
    b.record()
    torch.cuda.synchronize()
    print('weight gradient 256 -> 256 @ 28 x 28 alone: %.1f us per launch' % (a.elapsed_time(b) / 40 * 1e3))


if __name__ == '__main__':
    main()
