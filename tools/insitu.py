#!/usr/bin/env python3
"""Per-launch durations of the convolution launches of ZF_UNET's step IN SITU (both streams running, launch lists replayed
from C with the timing events inside), beside the stand-alone durations of tools/layer_bench.py's order.
    python tools/insitu.py [--batch 32 --size 224 --reps 8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from lib import losses as L
from lib.models.zf_unet import ZF_UNET
from segnb import engine, optim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--reps', type=int, default=8)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = ZF_UNET().to(dev).train()
    crit = L.BCEAndDiceLoss()
    opt = optim.SGD(model.parameters(), lr=1e-3)
    x = torch.randn(args.batch, 3, args.size, args.size).to(dev)
    y = (torch.rand(args.batch, 1, args.size, args.size) > 0.7).long().to(dev)

    def step():
        opt.zero_grad()
        loss = crit(model(x), y)
        (args.batch * loss).backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    timer = engine.KernelTimer()
    engine.TIMER = timer
    step()                      # records the lists with the events
    torch.cuda.synchronize()
    assert timer.persistent
    n = len(timer.records)
    acc = [0.0] * n
    for _ in range(args.reps):
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        for i, (label, flops, a, b) in enumerate(timer.records):
            acc[i] += a.elapsed_time(b) * 1e3
    engine.TIMER = None
    enc = ['enc%d.%s' % (i, l) for i in range(6) for l in ('l1', 'l2')]
    dec = ['dec%d.%s' % (i, l) for i in (4, 3, 2, 1, 0) for l in ('l1', 'l2')]
    fwd_names = enc + dec
    fl = [r for r in timer.records]
    names, k_f, k_w = [], 0, 0
    # forward launches come first (22), then backward in reverse layer order: data gradient (all but enc0.l1) and weight
    # gradient of each layer as the launcher issued them
    order_bwd = list(reversed(fwd_names))
    seen_f = 0
    bw_f = []
    for nm in order_bwd:
        if nm == 'enc0.l1':
            continue
        if nm.startswith('dec') and nm.endswith('l1'):
            lvl = int(nm[3])
            from lib.models.zf_unet import DECODER
            conv = model._engine.stages[DECODER[4 - lvl]][0].conv
            if hasattr(conv, 'segmented') and conv.segmented(args.batch, args.size >> lvl, args.size >> lvl):
                bw_f += [nm + ' skip', nm + ' up']      # the data gradient by segment: two launches
                continue
        bw_f.append(nm)
    for label, flops, a, b in fl:
        if label == 'conv_fprop':
            if seen_f < 22:
                names.append('fprop ' + fwd_names[seen_f])
            else:
                names.append('dgrad ' + (bw_f[seen_f - 22] if seen_f - 22 < len(bw_f) else '?'))
            seen_f += 1
        else:
            names.append('wgrad ' + order_bwd[k_w])
            k_w += 1
    tot = {}
    for i, nm in enumerate(names):
        us = acc[i] / args.reps
        kind = nm.split()[0]
        tot[kind] = tot.get(kind, 0.0) + us
        print('%-16s %8.1f us  %7.1f TF/s' % (nm, us, fl[i][1] / us / 1e6))
    print('totals (us): ' + '  '.join('%s %.0f' % kv for kv in tot.items()))


if __name__ == '__main__':
    main()
