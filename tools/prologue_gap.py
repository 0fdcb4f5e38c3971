#!/usr/bin/env python3
"""GPU time between the optimizer's kernel of step k and the first recorded launch of step k + 1's forward, measured with
HIP events on the launch stream over unsynchronised consecutive steps (no profiler attached): what rocprofv3's kernel trace
showed as a 0.18-0.27 ms hole at the start of a step (profiles/r04_step_trace.txt) -- tracer artefact or serial prologue?

    python tools/prologue_gap.py [--steps 40]
Prints per step: prologue (SGD end -> first launch of the forward list), forward list, rest (loss + backward + SGD)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    args = ap.parse_args()
    from lib.losses import BCEAndDiceLoss
    from lib.models import zf_unet as Z
    from segnb import optim as sopt
    torch.manual_seed(0)
    m = Z.ZF_UNET().set_compute_dtype('bf16').cuda().train()
    opt = sopt.SGD(m.parameters(), lr=1e-3)
    x = torch.randn(32, 3, 224, 224, device='cuda')
    y = (torch.rand(32, 1, 224, 224, device='cuda') > 0.5).long()
    crit = BCEAndDiceLoss()
    marks = []
    orig = Z._ZFUnetPlan._plan_replay
    state = {'fwd': True}

    def replay(self, plan, H=None, W=None):
        kind = 'fwd' if H is None else 'bwd'
        if kind == 'fwd':
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(('fwd_list_start', e))
        r = orig(self, plan, H, W)
        if kind == 'fwd':
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(('fwd_list_end', e))
        return r
    Z._ZFUnetPlan._plan_replay = replay

    def step():
        opt.zero_grad()
        out = m(x)
        loss = crit(out, y)
        (32 * loss).backward()
        opt.step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(('step_end', e))
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    del marks[:]
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ends = [e for k, e in marks if k == 'step_end']
    starts = [e for k, e in marks if k == 'fwd_list_start']
    fends = [e for k, e in marks if k == 'fwd_list_end']
    pro, fwd, rest, tot = [], [], [], []
    for i in range(1, len(ends)):
        pro.append(ends[i - 1].elapsed_time(starts[i]) * 1e3)
        fwd.append(starts[i].elapsed_time(fends[i]) * 1e3)
        rest.append(fends[i].elapsed_time(ends[i]) * 1e3)
        tot.append(ends[i - 1].elapsed_time(ends[i]) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print('steps %d (unsynchronised, three event records per step on the launch stream)' % len(tot))
    print('prologue  SGD end -> first launch of the forward list: median %.1f us  (min %.1f, max %.1f)' % (med(pro), min(pro), max(pro)))
    print('          = dropout tables (2 torch launches) + weight pack + input pack, incl. two event packets')
    print('forward list: median %.1f us;  head + loss + backward + unpack + SGD: median %.1f us;  step: median %.1f us'
          % (med(fwd), med(rest), med(tot)))
    print('last three steps, prologue us: ' + ' '.join('%.1f' % v for v in pro[-3:]))


if __name__ == '__main__':
    main()
