#!/usr/bin/env python3
"""Thread sweep of bench.py's CPU baseline (oracle/train_step_ref.train_step, ZF_UNET fp32 B=4 224x224) on this box's host
cores: images/s at 8 / 16 / 32 / 64 / 128 / 256 threads (those the box has), ~8 s each.  Justifies the thread count
bench.py's cpu_baseline() uses.    python tools/cpu_sweep.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch


def main():
    from oracle import train_step_ref, zf_unet_ref
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    B, S = 4, 224
    x, y = train_step_ref.synthetic_batch(B, S, seed=1234)
    sd = zf_unet_ref.default_init_state(filters=32, seed=0)
    out = {'host_cores': os.cpu_count(), 'affinity_cores': ncpu, 'torch': torch.__version__, 'sweep': []}
    for nt in (8, 16, 32, 64, 128, 256):
        if nt > ncpu:
            break
        torch.set_num_threads(nt)
        train_step_ref.train_step(sd, x, y, 'bce_dice', lr=1e-3)
        t0 = time.time()
        n = 0
        while n < 3 or (time.time() - t0 < 8.0 and n < 20):
            train_step_ref.train_step(sd, x, y, 'bce_dice', lr=1e-3)
            n += 1
        dt = time.time() - t0
        out['sweep'].append({'threads': nt, 'images_per_s': round(B * n / dt, 2), 'steps': n})
        print(json.dumps(out['sweep'][-1]), flush=True)
    best = max(out['sweep'], key=lambda r: r['images_per_s'])
    out['best'] = best
    print(json.dumps(out))


if __name__ == '__main__':
    main()
