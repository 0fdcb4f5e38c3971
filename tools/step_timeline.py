#!/usr/bin/env python3
"""Where one training step of the headline configuration spends its time on the two streams (HIP events, no profiler):
forward end, end of the main stream's backward chain (the moment it starts waiting for the weight-gradient stream), end
of the weight-gradient stream, end of the step.     python tools/step_timeline.py [--steps 20]"""
import argparse
import os
import sys

os.environ['SEGNB_CPLAN'] = '0'      # the marks hook the Python launcher (join_side); the replayed lists issue the same launches

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from lib import losses as L
from lib.models.zf_unet import ZF_UNET
from segnb import optim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = ZF_UNET().to(dev).train()
    crit = L.BCEAndDiceLoss()
    opt = optim.SGD(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)
    y = (torch.rand(args.batch, 1, args.size, args.size, generator=g) > 0.7).long().to(dev)
    marks = {}

    def ev(stream=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        return e

    def step(record):
        if record:
            marks['t0'] = ev()
        opt.zero_grad()
        out = model(x)
        if record:
            marks['fwd'] = ev()
        loss = crit(out, y)
        (x.size(0) * loss).backward()
        if record:
            marks['bwd'] = ev()
        opt.step()
        if record:
            marks['end'] = ev()

    for _ in range(5):
        step(False)
    rt = model._engine.rt
    orig_join = rt.join_side

    def join_side():
        if getattr(rt, '_side_busy', False) and 'rec' in marks:
            marks['main_chain'] = ev()
            marks['side_done'] = ev(rt._side)
        orig_join()
    rt.join_side = join_side
    acc = {}
    for _ in range(args.steps):
        marks.clear()
        marks['rec'] = True
        step(True)
        torch.cuda.synchronize()
        for k in ('fwd', 'main_chain', 'side_done', 'bwd', 'end'):
            acc[k] = acc.get(k, 0.0) + marks['t0'].elapsed_time(marks[k])
    for k in ('fwd', 'main_chain', 'side_done', 'bwd', 'end'):
        print('%-12s %7.3f ms after the step started' % (k, acc[k] / args.steps))
    print('(synchronised every step: the host enqueue of a step is not hidden behind the previous one here)')


if __name__ == '__main__':
    main()
