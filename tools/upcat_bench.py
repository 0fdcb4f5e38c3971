"""Per-layer timing of the decoder blocks' first convolution (cat([Upsample x2(u), skip]) -> conv3x3) at the timed
configuration (bs=32): the plain 9-tap launches over the concat buffer (ConvOp) against the segmented form
(UpCatConvOp: skip segment + the upsampled segment on the low-resolution tensor).  Launches alone, HIP events.

    python tools/upcat_bench.py [--reps 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'segmentation-networks-benchmark_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

from segnb import _native as nv
from segnb import convplan as cp
from segnb.engine import ConvOp, PackTable, Runtime, UpCatConvOp, View

SHAPES = [('dec4.l1', 14, 1024, 512, 512), ('dec3.l1', 28, 512, 256, 256), ('dec2.l1', 56, 256, 128, 128),
          ('dec1.l1', 112, 128, 64, 64), ('dec0.l1', 224, 64, 32, 32)]


def timed(fn, reps):
    for _ in range(3):
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
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    nv.load()
    rt = Runtime('cuda', 'bf16')
    N = args.batch
    tot = {}
    for name, S, Cu, Cs, Co in SHAPES:
        w = torch.randn(Co, Cu + Cs, 3, 3, device='cuda') * (2.0 / ((Cu + Cs) * 9)) ** 0.5
        plain = ConvOp(rt, w, None, [(Cu, Cu), (Cs, Cs)], 1, 1, False, True)
        seg = UpCatConvOp(rt, w, None, [(Cu, Cu), (Cs, Cs)], True)
        PackTable(rt, plain.pack_jobs(S, S) + seg.pack_jobs(S, S), 'segnb_pack_weight_multi', 'segnb_pack_weight').run()
        cat = View.alloc(rt, N, S, S, Cu + Cs)
        cat.t.normal_()
        u = View.alloc(rt, N, S // 2, S // 2, Cu)
        u.t.normal_()
        du = View.alloc(rt, N, S // 2, S // 2, Cu)
        dcat = View.alloc(rt, N, S, S, Cu + Cs)
        dy = View.alloc(rt, N, S, S, Co)
        dy.t.normal_()
        y = View.alloc(rt, N, S, S, Co)
        gw = torch.zeros_like(w)
        seg.bind_up(u, du)
        seg.segment_fwd = True
        seg.virtual_concat, seg.segment_wgrad = True, False
        virt = seg.virtual(N, S, S)
        tv = (timed(lambda: seg.fprop(cat, y), args.reps), timed(lambda: seg.wgrad(cat, dy, gw, unpack=False), args.reps)) if virt else (float('nan'),) * 2
        seg.virtual_concat = False
        seg._seg.clear()
        fseg = seg.fwd_segmented(N, S, S, Co)
        res = {
            'fprop 9-tap': timed(lambda: plain.fprop(cat, y), args.reps),
            'fprop virt': tv[0], 'wgrad virt': tv[1],
            'fprop skip': timed(lambda: seg.skip.fprop(cat.slice(Cu, Cs), y, None), args.reps),
            'fprop up': timed(lambda: seg.up.fprop_acc(u, y, None), args.reps) if fseg else float('nan'),
            'dgrad 9-tap': timed(lambda: plain.dgrad(dy, dcat), args.reps),
            'dgrad skip': timed(lambda: seg.skip.dgrad(dy, dcat.slice(Cu, Cs)), args.reps),
            'dgrad up': timed(lambda: seg.up.dgrad(dy, du), args.reps),
            'wgrad 9-tap': timed(lambda: plain.wgrad(cat, dy, gw, unpack=False), args.reps),
            'wgrad skip': timed(lambda: seg.skip.wgrad(cat.slice(Cu, Cs), dy, gw, unpack=False), args.reps),
            'wgrad up': timed(lambda: seg.up.wgrad(u, dy, gw, unpack=False), args.reps),
        }
        print('%-8s %4dx%-4d %4d+%-4d->%-4d ' % (name, S, S, Cu, Cs, Co) +
              '  '.join('%s %6.1f' % (k, v) for k, v in res.items()))
        for k, v in res.items():
            tot[k] = tot.get(k, 0.0) + v
    print('totals (us): ' + '  '.join('%s %6.1f' % (k, v) for k, v in tot.items()))
    print('dgrad: 9-tap %.0f -> segmented %.0f us;  wgrad: 9-tap %.0f -> segmented %.0f us'
          % (tot['dgrad 9-tap'], tot['dgrad skip'] + tot['dgrad up'], tot['wgrad 9-tap'], tot['wgrad skip'] + tot['wgrad up']))


if __name__ == '__main__':
    main()
