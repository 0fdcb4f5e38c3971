#!/usr/bin/env python3
"""First-layer weight gradient (3 -> 32 at bs=32 224x224) stand-alone: the plain kernel on a stored dz, the apply pass that
stores it, and the variant that recomputes dz from (g, y) (segnb_conv_wgrad_bnapply).  HIP events, mean of --reps launches.

    python tools/c8_bench.py [--reps 30] [--c8roll 0|1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
os.environ['SEGNB_WGRAD_BNAPPLY'] = '1'
import torch

from segnb import _native as nv
from segnb import convplan as cp
from segnb.engine import ConvOp, Runtime, View


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--c8roll', type=int, default=1, help='segnb_tune wgrad_c8roll')
    args = ap.parse_args()
    nv.call('segnb_tune', b'wgrad_c8roll', args.c8roll)
    rt = Runtime('cuda', 'bf16')
    N, H, W, Ci, Co = args.batch, args.size, args.size, 3, 32
    w = torch.randn(Co, Ci, 3, 3).cuda()
    op = ConvOp(rt, w, None, [(Ci, cp.pad8(Ci))], 1, 1, False, need_dgrad=False)
    xv = View.alloc(rt, N, H, W, op.Cip)
    xv.dense()[..., :Ci].normal_()
    yv, gv, dz = (View.alloc(rt, N, H, W, op.Cop) for _ in range(3))
    yv.t.normal_()
    gv.t.normal_()
    Cp = op.Cop
    coef = torch.stack([0.5 + torch.rand(Cp), 0.3 * torch.randn(Cp), 0.2 * torch.randn(Cp), 0.5 + torch.rand(Cp)]).cuda()
    bcoef = torch.stack([0.5 + torch.rand(Cp), 0.1 * torch.randn(Cp), 0.1 * torch.randn(Cp)]).cuda()
    gw = torch.zeros_like(w)
    fns = {
        'apply pass (g, y -> dz)': lambda: nv.call('segnb_bn_bwd_apply_direct', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef),
                                                   nv.ptr(bcoef), nv.ACT_RELU, 0.0, gv.ptr, gv.ld, dz.ptr, dz.ld, None, Co, rt.stream),
        'wgrad on stored dz': lambda: op.wgrad(xv, dz, gw, unpack=False),
        'wgrad recomputing dz': lambda: op.wgrad_bnapply(xv, gv, yv, coef, bcoef, nv.ACT_RELU, 0.0),
    }
    for name, fn in fns.items():
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        print('%-26s %7.1f us' % (name, a.elapsed_time(b) / args.reps * 1e3))


if __name__ == '__main__':
    main()
