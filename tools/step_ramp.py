#!/usr/bin/env python3
"""Per-step GPU time of the first steps after an idle period (HIP events around every step, no host sync inside): does a 20-step
timed region behind 5 warm-up steps run at the steady-state clock?   python tools/step_ramp.py [--idle-ms 500] [--steps 80]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=80)
    ap.add_argument('--idle-ms', type=float, default=500.0)
    args = ap.parse_args()
    from lib.losses import BCEAndDiceLoss
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(0)
    m = ZF_UNET().set_compute_dtype('bf16').cuda().train()
    from segnb import optim as sopt
    opt = sopt.SGD(m.parameters(), lr=1e-3)
    x = torch.randn(32, 3, 224, 224, device='cuda')
    y = (torch.rand(32, 1, 224, 224, device='cuda') > 0.5).long()
    crit = BCEAndDiceLoss()

    def step():
        opt.zero_grad()
        out = m(x)
        loss = crit(out, y)
        (32 * loss).backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for rnd in range(2):
        time.sleep(args.idle_ms / 1e3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        ev[0].record()
        for i in range(args.steps):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
        print('round %d after %.0f ms idle: steps 1-5 %s | 6-25 mean %.3f | 26-45 mean %.3f | last 20 mean %.3f' % (
            rnd, args.idle_ms, ' '.join('%.2f' % v for v in ms[:5]), sum(ms[5:25]) / 20, sum(ms[25:45]) / 20, sum(ms[-20:]) / 20))


if __name__ == '__main__':
    main()
