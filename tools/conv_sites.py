#!/usr/bin/env python3
"""Per-convolution timing of one training step of a model (every launch synchronised: stand-alone durations).

    python tools/conv_sites.py --model linknet34 [--batch 16 --size 512]

For each ConvOp call: kind (fprop / dgrad / wgrad), geometry, microseconds, algorithmic TFLOP/s; then the totals per kind
and the slowest sites.  One stream (SEGNB_OVERLAP_WGRAD=0) so that nothing runs beside the measured launch."""
import argparse
import os
import sys
import warnings

os.environ.setdefault('SEGNB_OVERLAP_WGRAD', '0')
os.environ['SEGNB_CPLAN'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='linknet34')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--top', type=int, default=40)
    args = ap.parse_args()
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.linknet import LinkNet34
    from lib.models.tiramisu import FCDenseNet103, FCDenseNet67
    from lib.models.unet16 import UNet16
    from segnb import engine as E, optim
    ctor, B, S = {'linknet34': (LinkNet34, 16, 512), 'unet16': (UNet16, 4, 512),
                  'fcdensenet103': (lambda: FCDenseNet103(n_classes=1), 8, 256),
                  'fcdensenet67': (lambda: FCDenseNet67(n_classes=1), 8, 256)}[args.model]
    B, S = args.batch or B, args.size or S
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ctor().cuda().train()
    x = torch.randn(B, 3, S, S).cuda()
    y = (torch.rand(B, 1, S, S) > 0.7).long().cuda()
    crit = BCEWithLogitsLossAndSmoothJaccard()
    opt = optim.SGD(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad()
        loss = crit(m(x), y)
        (B * loss).backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rows = []

    def wrap(name):
        fn = getattr(E.ConvOp, name)

        def timed(self, a, b, *rest, **kw):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(self, a, b, *rest, **kw)
            e1.record()
            torch.cuda.synchronize()
            xin = a if name != 'dgrad' else b          # the layer's INPUT view (dgrad: a = dy, b = dx)
            p = self.plan(xin.H, xin.W)
            Ho, Wo = p['out_hw']
            ntaps = sum(len(l.taps) * l.QH * l.QW for l in p['fwd']) if not self.transposed else \
                sum(len(l.taps) * l.QH * l.QW for l in p['dg'])
            flops = 2.0 * xin.N * ntaps * self.Ci * self.Co
            k = int(round((self.weight.shape[2] * self.weight.shape[3]) ** 0.5))
            rows.append((name, '%s%dx%d s%d %d->%d in %dx%d out %dx%d' % ('T' if self.transposed else '', k, k, self.stride,
                                                                          self.Ci, self.Co, xin.H, xin.W, Ho, Wo),
                         e0.elapsed_time(e1) * 1e3, flops))
            return out
        setattr(E.ConvOp, name, timed)
    for n in ('fprop', 'dgrad', 'wgrad'):
        wrap(n)
    step()
    tot = {}
    agg = {}
    for kind, geom, us, fl in rows:
        tot[kind] = tot.get(kind, 0.0) + us
        a = agg.setdefault((kind, geom), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += us
        a[2] += fl
    print('%s B=%d %dx%d: per-kind totals (us): %s' % (args.model, B, S, S, '  '.join('%s %.0f' % kv for kv in tot.items())))
    buckets = {}
    for kind, geom, us, fl in rows:
        size = geom.split(' in ')[1].split(' ')[0]
        b = buckets.setdefault((kind, geom.split(' ')[0] + ' ' + geom.split(' ')[1], size), [0, 0.0])
        b[0] += 1
        b[1] += us
    print('buckets (kind, kernel, input size): launches, total us')
    for k, (n, us) in sorted(buckets.items(), key=lambda kv: -kv[1][1])[:30]:
        print('  %-6s %-10s %-9s x%-3d %8.1f us' % (k[0], k[1], k[2], n, us))
    for (kind, geom), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print('%-6s %-46s x%-3d %8.1f us total %8.1f us each %7.1f TF/s' % (kind, geom, n, us, us / n, fl / us / 1e6))


if __name__ == '__main__':
    main()
