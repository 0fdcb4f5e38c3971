"""Run-to-run reproducibility of a training step: the same batch, the same parameters, the same Dropout2d seed, K times in one process;
how many DIFFERENT logits / gradients come out (1 = bitwise reproducible), eager launcher and replayed launch lists.

    python tools/determinism_probe.py [--steps 8]
"""
import argparse
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'segmentation-networks-benchmark_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

from lib.losses import BCEWithLogitsLossAndSmoothJaccard


def models():
    from lib.models.zf_unet import ZF_UNET
    from lib.models.linknet import LinkNet34
    from lib.models.tiramisu import FCDenseNet67
    from lib.models.unet16 import UNet16
    return [('ZF_UNET', lambda: ZF_UNET(), (4, 3, 224, 224)), ('LinkNet34', lambda: LinkNet34(), (4, 3, 256, 256)),
            ('FCDenseNet67', lambda: FCDenseNet67(n_classes=1), (2, 3, 128, 128)), ('UNet16', lambda: UNet16(), (2, 3, 256, 256))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    args = ap.parse_args()
    crit = BCEWithLogitsLossAndSmoothJaccard()
    for name, ctor, shape in models():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            torch.manual_seed(3)
            m = ctor().cuda().train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(*shape, generator=g).cuda()
        y = (torch.rand(shape[0], 1, shape[2], shape[3], generator=g) > 0.7).long().cuda()
        outs, grads = [], []
        for i in range(args.steps):
            torch.manual_seed(11)
            m.zero_grad()
            o = m(x)
            loss = crit(o, y)
            (shape[0] * loss).backward()
            torch.cuda.synchronize()
            outs.append(o.detach().clone())
            grads.append(torch.cat([p.grad.detach().reshape(-1) for p in m.parameters()]).clone())

        def distinct(ts):
            reps = []
            for t_ in ts:
                if not any(torch.equal(t_, r) for r in reps):
                    reps.append(t_)
            return reps
        # steps 0 / 1 run the eager launcher (1 records the lists), 2.. replay them
        do, dg = distinct(outs), distinct(grads)
        rel = max(float((a - dg[0]).norm() / (dg[0].norm() + 1e-30)) for a in dg)
        rg = distinct(grads[2:])
        rel_r = max(float((a - rg[0]).norm() / (rg[0].norm() + 1e-30)) for a in rg)
        first = 'differs from' if not torch.equal(outs[0], outs[2]) else 'equals'
        print('%-14s %d steps: %d distinct logits, %d distinct gradients (largest relative distance from the first %.2e); replayed steps '
              'only: %d / %d (%.2e); step 0 %s step 2 in its logits' % (name, args.steps, len(do), len(dg), rel, len(distinct(outs[2:])),
                                                                      len(rg), rel_r, first))


if __name__ == '__main__':
    main()
