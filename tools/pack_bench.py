#!/usr/bin/env python3
"""The weight pack at the start of a ZF_UNET step (fp32 parameters -> bf16 matrices of the forward and the data gradients),
split by job class: where do the ~100 us go?  HIP events, mean of --reps launches.

    python tools/pack_bench.py [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb.engine import PackTable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    args = ap.parse_args()
    from lib.models.zf_unet import ZF_UNET
    m = ZF_UNET().cuda().train()
    eng = m._get_engine(torch.device('cuda', 0))
    eng.rt.set_dtype('bf16') if hasattr(eng.rt, 'set_dtype') else None
    N, H, W = args.batch, args.size, args.size
    x = torch.randn(N, 3, H, W, device='cuda')
    with torch.autocast('cuda', dtype=torch.bfloat16):
        m(x)
    torch.cuda.synchronize()
    jobs = []
    for conv, h, w in eng._conv_sizes(H, W):
        jobs += conv.pack_jobs(h, w, N) if hasattr(conv, 'segmented') else conv.pack_jobs(h, w)

    def nbytes(js):
        rd = sum(j['w'].numel() * 4 for j in js)
        wr = sum(j['packed'].numel() * j['packed'].element_size() for j in js)
        return rd, wr
    classes = {
        'all': jobs,
        'forward forms': [j for j in jobs if j.get('form') == 'f'],
        'data-gradient forms': [j for j in jobs if j.get('form') == 'd'],
        'plain (not masked)': [j for j in jobs if not j.get('masked')],
        'masked (upsampled segments)': [j for j in jobs if j.get('masked')],
        'plain forward': [j for j in jobs if not j.get('masked') and j.get('form') == 'f'],
        'plain data-gradient': [j for j in jobs if not j.get('masked') and j.get('form') == 'd'],
    }
    for pair, (name, js) in [(pr, it) for pr in (False, True) for it in classes.items()]:
        PackTable.pair_pack = pair
        if pair and name not in ('all', 'plain (not masked)'):
            continue
        name = name + (' PAIRED' if pair else '')
        t = PackTable(eng.rt, js, 'segnb_pack_weight_multi', 'segnb_pack_weight')
        t.run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            t.run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / args.reps * 1e3
        rd, wr = nbytes(js)
        if pair and name.startswith('all'):
            for j in js:
                if id(j) not in t.paired_ids:
                    print('   unpaired: form %s masked %s Mp %d Cp %d ntaps %d s_m %d s_c %d' % (
                        j.get('form'), bool(j.get('masked')), j['Mp'], j['Cp'], j['ntaps'], j['s_m'], j['s_c']))
        print('%-30s %3d jobs %6d + %5d blocks  %7.1f us   read %6.1f MB  write %6.1f MB' % (
            name, len(js), t.blocks, t.pblocks, us, rd / 1e6, wr / 1e6))


if __name__ == '__main__':
    main()
