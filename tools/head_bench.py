#!/usr/bin/env python3
"""Stand-alone timing of the classifier-head kernels (segnb_head_conv_fwd / _bwd, segnb_head_fwd / _bwd) at the models' shapes.

    python tools/head_bench.py            # LinkNet34 finalconv3 (16 x 511^2 x 32, 2 x 2 window) and UNet16 final (4 x 1024^2 x 32)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb import _native as nv
from segnb.engine import Runtime, View


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
    for name, (N, H, W, C, K, kh, kw, pad) in (('linknet34 finalconv3', (16, 511, 511, 32, 1, 2, 2, 1)),
                                               ('unet16 final', (4, 1024, 1024, 32, 1, 1, 1, 0))):
        Ho, Wo = H + 2 * pad - kh + 1, W + 2 * pad - kw + 1
        av = View.alloc(rt, N, H, W, C)
        av.dense().normal_()
        da = View.alloc(rt, N, H, W, C)
        w = torch.randn(K, C, kh, kw, device='cuda') * 0.1
        b = torch.zeros(K, device='cuda')
        logits = torch.zeros(N, K, Ho, Wo, device='cuda')
        dl = torch.randn(N, K, Ho, Wo, device='cuda')
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        sums = torch.zeros(16, 2, C, dtype=torch.float64, device='cuda')
        bytes_f = N * H * W * C * 2 + logits.numel() * 4
        bytes_b = 2 * N * H * W * C * 2 + logits.numel() * 4
        tf = timed(lambda: nv.call('segnb_head_conv_fwd', rt.code, av.ptr, av.ld, N, H, W, C, nv.ptr(w), kh, kw, pad, nv.ptr(b), K,
                                   nv.ptr(logits), rt.stream))
        print('%-22s head_conv_fwd          %7.1f us  %5.2f TB/s' % (name, tf, bytes_f / tf / 1e6))
        for act in (-1, nv.ACT_LEAKY):
            tb = timed(lambda: nv.call('segnb_head_conv_bwd', rt.code, av.ptr, av.ld, N, H, W, C, C, nv.ptr(w), kh, kw, pad, K,
                                       nv.ptr(dl), act, 0.01, da.ptr, da.ld, nv.ptr(dw), nv.ptr(db),
                                       nv.ptr(sums) if act >= 0 else None, rt.stream))
            print('%-22s head_conv_bwd act=%2d   %7.1f us  %5.2f TB/s' % (name, act, tb, bytes_b / tb / 1e6))
        if kh == 1:
            tf = timed(lambda: nv.call('segnb_head_fwd', rt.code, av.ptr, av.ld, N, H, W, C, nv.ptr(w), nv.ptr(b), K, nv.ptr(logits),
                                       rt.stream))
            tb = timed(lambda: nv.call('segnb_head_bwd', rt.code, av.ptr, av.ld, N, H, W, C, C, nv.ptr(w), K, nv.ptr(dl), da.ptr, da.ld,
                                       nv.ptr(dw), nv.ptr(db), rt.stream))
            print('%-22s head_fwd / head_bwd    %7.1f / %.1f us' % (name, tf, tb))


if __name__ == '__main__':
    main()
