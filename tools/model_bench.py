#!/usr/bin/env python3
"""Training-step throughput of every model family at the SURVEY 8d configurations (synthetic data, bf16, SGD,
bce_jaccard): not the headline metric (bench.py), a record that the other rows run at their real sizes.
    python tools/model_bench.py [--steps 10]"""
import argparse
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from lib.losses import BCEWithLogitsLossAndSmoothJaccard
from segnb import optim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    from lib.models.zf_unet import ZF_UNET
    from lib.models.linknet import LinkNet34
    from lib.models.tiramisu import FCDenseNet103, FCDenseNet67
    from lib.models.unet16 import UNet16
    cfgs = [('ZF_UNET 512x512 bs=16 train', ZF_UNET, 16, 512, True, 414.1),
            ('LinkNet34 512x512 bs=16 train', LinkNet34, 16, 512, True, 138.5),
            ('FCDenseNet103 256x256 bs=8 train', lambda: FCDenseNet103(n_classes=1), 8, 256, True, 156.1),
            ('FCDenseNet67 256x256 bs=8 train', lambda: FCDenseNet67(n_classes=1), 8, 256, True, 6 * 27.11),
            ('UNet16 1024x1024 bs=4 eval forward', UNet16, 4, 1024, False, 1279.0)]
    for name, ctor, B, S, train, gflop in cfgs:
        if args.only and args.only not in name:
            continue
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = ctor().cuda()
        x = torch.randn(B, 3, S, S).cuda()
        y = (torch.rand(B, 1, S, S) > 0.7).long().cuda()
        crit = BCEWithLogitsLossAndSmoothJaccard()
        if train:
            m.train()
            opt = optim.SGD(m.parameters(), lr=1e-3)

            def step():
                opt.zero_grad()
                loss = crit(m(x), y)
                (B * loss).backward()
                opt.step()
                return loss
        else:
            m.eval()

            def step():
                with torch.no_grad():
                    return m(x).mean()
        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print('%-36s %8.1f ms/step %8.1f img/s  %6.1f TFLOP/s algorithmic  (last value %.4f, peak mem %.1f GB)' % (
            name, dt * 1e3, B / dt, B / dt * gflop / 1e3, float(out), torch.cuda.max_memory_allocated() / 2**30))
        del m, x, y
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == '__main__':
    main()
