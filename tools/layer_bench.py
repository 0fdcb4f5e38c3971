#!/usr/bin/env python3
"""Per-layer timing of the convolution kernels on the ZF_UNET bs=32 224x224 shapes (kernel tuning aid).

    python tools/layer_bench.py [--reps 20] [--batch 32] [--size 224] [--what fprop,dgrad,wgrad]

Each line: layer, shape, time per launch (HIP events on the launch stream, mean of `reps` back-to-back
launches) and algorithmic TFLOP/s.  SEGNB_FPROP_GENERAL=1 / SEGNB_WGRAD_GENERAL=1 select the general gather
kernels for an A/B in a second process."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb import convplan as cp
from segnb.engine import ConvOp, Runtime, View


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--what', default='fprop,dgrad,wgrad')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--dma', type=int, default=None, help='segnb_tune fprop_dma (0/1)')
    ap.add_argument('--cfg', type=int, default=None, help='segnb_tune fprop_dma_cfg')
    ap.add_argument('--dbg', type=int, default=None, help='segnb_tune fprop_dma_dbg (timing builds)')
    ap.add_argument('--rw', type=int, default=None, help='segnb_tune fprop_rw (0/1)')
    ap.add_argument('--mf16', type=int, default=None, help='segnb_tune fprop_mf16 (0/1)')
    ap.add_argument('--rwsw', type=int, default=None, help='segnb_tune rw_store_waves (2/4)')
    ap.add_argument('--nostats', type=int, default=None, help='segnb_tune fprop_nostats (0/1)')
    ap.add_argument('--roll', type=int, default=None, help='segnb_tune fprop_roll (0/1/2)')
    ap.add_argument('--wroll', type=int, default=None, help='segnb_tune wgrad_roll (0/1)')
    ap.add_argument('--wg-all', type=int, default=1,
                    help='1 (default): a second weight-gradient column with the pixel split sized for ALL CUs (segnb_tune wg_cu_pct 100); '
                         'the first column is the training step\'s configuration -- the wide launches sized for half of the CUs, because '
                         'they run beside the data-gradient chain')
    ap.add_argument('--ksplit', type=int, default=None, help='segnb_tune fprop_ksplit (0 off / 1 auto / 2 / 4)')
    ap.add_argument('--only', default='', help='comma-separated layer names')
    ap.add_argument('--wgrad-unpack', type=int, default=0,
                    help='1: time the per-layer unpack (packed fp32 workspace -> parameter-layout gradient) with the '
                         'weight gradient; the training step unpacks all layers in three batched launches instead')
    args = ap.parse_args()
    rt = Runtime('cuda', args.dtype)
    from segnb import _native as nv
    if args.dma is not None:
        nv.call('segnb_tune', b'fprop_dma', args.dma)
    if args.dbg is not None:
        nv.call('segnb_tune', b'fprop_dma_dbg', args.dbg)
    if args.nostats is not None:
        nv.call('segnb_tune', b'fprop_nostats', args.nostats)
    if args.rwsw is not None:
        nv.call('segnb_tune', b'rw_store_waves', args.rwsw)
    if args.mf16 is not None:
        nv.call('segnb_tune', b'fprop_mf16', args.mf16)
    if args.rw is not None:
        nv.call('segnb_tune', b'fprop_rw', args.rw)
    if args.roll is not None:
        nv.call('segnb_tune', b'fprop_roll', args.roll)
    if args.wroll is not None:
        nv.call('segnb_tune', b'wgrad_roll', args.wroll)
    if args.cfg is not None:
        nv.call('segnb_tune', b'fprop_dma_cfg', args.cfg)
    if args.ksplit is not None:
        nv.call('segnb_tune', b'fprop_ksplit', args.ksplit)
    f, N, S = 32, args.batch, args.size
    w = [f, 2 * f, 4 * f, 8 * f, 16 * f, 32 * f]
    layers = []
    cin = 3
    for i in range(6):
        layers.append(('enc%d.l1' % i, S >> i, [(cin, cp.pad8(cin))], w[i]))
        layers.append(('enc%d.l2' % i, S >> i, [(w[i], w[i])], w[i]))
        cin = w[i]
    for lvl in (4, 3, 2, 1, 0):
        layers.append(('dec%d.l1' % lvl, S >> lvl, [(w[lvl + 1], w[lvl + 1]), (w[lvl], w[lvl])], w[lvl]))
        layers.append(('dec%d.l2' % lvl, S >> lvl, [(w[lvl], w[lvl])], w[lvl]))
    tot = {}
    only = [s for s in args.only.split(',') if s]
    for name, hw, segs, co in layers:
        if only and name not in only:
            continue
        ci = sum(r for r, _ in segs)
        wt = torch.randn(co, ci, 3, 3, device='cuda') * 0.05
        op = ConvOp(rt, wt, torch.zeros(co, device='cuda'), segs, 1, 1, False, True)
        op.pack(hw, hw)
        xv = View.alloc(rt, N, hw, hw, op.Cip)
        xv.t.normal_()
        yv = View.alloc(rt, N, hw, hw, op.Cop)
        dyv = View.alloc(rt, N, hw, hw, op.Cop)
        dyv.t.normal_()
        dxv = View.alloc(rt, N, hw, hw, op.Cip)
        gw = torch.zeros_like(wt)
        stats = rt.zeros((16, 2, op.Cop), torch.float64)
        flops = 2.0 * N * hw * hw * 9 * ci * co
        line = '%-9s %4dx%-4d %4d->%-4d' % (name, hw, hw, ci, co)
        whats = args.what.split(',')
        if args.wg_all and 'wgrad' in whats:
            whats = whats + ['wgrad@allCUs']
        for what in whats:
            if what == 'wgrad@allCUs':
                # the same layer with its pixel split (and workspace) planned for every CU
                nv.call('segnb_tune', b'wg_cu_pct', 100)
                op2 = ConvOp(rt, wt, torch.zeros(co, device='cuda'), segs, 1, 1, False, True)
                op2.pack(hw, hw)
                fn = lambda: op2.wgrad(xv, dyv, gw, unpack=bool(args.wgrad_unpack))
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.reps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                nv.call('segnb_tune', b'wg_cu_pct', 0)
                us = a.elapsed_time(b) / args.reps * 1e3
                tot[what] = tot.get(what, 0.0) + us
                line += '  %s %7.1f us %6.0f TF' % (what, us, flops / us / 1e6)
                continue
            fn = {'fprop': lambda: op.fprop(xv, yv, stats), 'dgrad': lambda: op.dgrad(dyv, dxv),
                  'wgrad': lambda: op.wgrad(xv, dyv, gw, unpack=bool(args.wgrad_unpack))}[what]
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / args.reps * 1e3
            tot[what] = tot.get(what, 0.0) + us
            line += '  %s %7.1f us %6.0f TF' % (what, us, flops / us / 1e6)
        print(line)
    print('totals (us): ' + '  '.join('%s %.0f' % kv for kv in tot.items()))


if __name__ == '__main__':
    main()
