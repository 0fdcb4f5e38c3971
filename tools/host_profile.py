"""Where the HOST spends its time enqueuing one ZF_UNET train step (no synchronisation inside the step).

    python tools/host_profile.py [--batch 32] [--size 224] [--cprofile]

Prints the per-segment host time (zero_grad / forward / loss / backward / optimizer) averaged over a few steps that start
from an idle GPU, with the C-side launch plans (SEGNB_CPLAN) on and off, and optionally the cProfile top list."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--cprofile', action='store_true')
    ap.add_argument('--methods', action='store_true', help='host time per engine method (inclusive)')
    ap.add_argument('--plan-profile', action='store_true', help='per-entry-point host time inside segnb_plan_run')
    args = ap.parse_args()
    from lib import losses as L
    from lib.models import zf_unet as zf
    from segnb import optim
    dev = torch.device('cuda:0')
    B, S = args.batch, args.size
    x = torch.randn(B, 3, S, S).to(dev)
    y = (torch.rand(B, 1, S, S) > 0.7).long().to(dev)
    for mode in (True, False):
        zf._ZFUnetPlan.use_cplan = mode
        torch.manual_seed(0)
        model = zf.ZF_UNET().set_compute_dtype('bf16').to(dev).train()
        crit = L.BCEAndDiceLoss()
        opt = optim.SGD(model.parameters(), lr=1e-3)
        seg = dict(zero=0.0, fwd=0.0, loss=0.0, bwd=0.0, opt=0.0)

        def step(acc=None):
            t = [time.perf_counter()]
            opt.zero_grad(); t.append(time.perf_counter())
            out = model(x); t.append(time.perf_counter())
            loss = crit(out, y); t.append(time.perf_counter())
            (B * loss).backward(); t.append(time.perf_counter())
            opt.step(); t.append(time.perf_counter())
            if acc is not None:
                for k, a, b in zip(acc, t[:-1], t[1:]):
                    acc[k] += b - a
        for _ in range(6):
            step()
        reps = 8
        for _ in range(reps):
            torch.cuda.synchronize()
            for _ in range(args.steps):
                step(seg)
        torch.cuda.synchronize()
        n = reps * args.steps
        print('cplan=%d  B=%d %dx%d  host ms/step: ' % (mode, B, S, S)
              + '  '.join('%s %.3f' % (k, v / n * 1e3) for k, v in seg.items())
              + '  total %.3f' % (sum(seg.values()) / n * 1e3), flush=True)
        if mode and args.plan_profile:
            from segnb import _native as nv
            nv.call('segnb_tune', b'plan_profile', 1)
            torch.cuda.synchronize()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            nv.call('segnb_tune', b'plan_profile', 2)
            nv.call('segnb_tune', b'plan_profile', 0)
            print('(per %d steps)' % args.steps, flush=True)
        if mode and args.methods:
            from segnb import engine as E, _native as nv2
            acc = {}

            def wrap(owner, name, label=None):
                fn = getattr(owner, name)

                def timed(*a, **k):
                    t0 = time.perf_counter()
                    try:
                        return fn(*a, **k)
                    finally:
                        key = label or name
                        if key == 'nv.call':
                            key = 'nv.call ' + str(a[0])
                        e = acc.setdefault(key, [0, 0.0])
                        e[0] += 1
                        e[1] += time.perf_counter() - t0
                setattr(owner, name, timed)
                return fn
            eng = model._engine
            saved = []
            for owner, name in ((E.FlatParams, 'ensure'), (E.FlatParams, 'begin_backward'), (E.FlatParams, 'publish_grads'),
                                (type(eng), '_pack_if_needed'), (type(eng), '_dropout_tables'), (type(eng), '_cplan_key'),
                                (type(eng), '_after_backward'), (type(eng), 'forward'),
                                (type(eng), 'backward'), (type(eng), 'buffers'), (zf.ZF_UNET, 'forward')):
                if hasattr(owner, name):
                    lab = owner.__name__ + '.' + name
                    saved.append((owner, name, wrap(owner, name, lab)))
            saved.append((nv2, 'call', wrap(nv2, 'call', 'nv.call')))
            zf.nv.call = nv2.call
            E.nv.call = nv2.call
            torch.cuda.synchronize()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            for owner, name, fn in saved:
                setattr(owner, name, fn)
            for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
                print('  %-44s %4d calls %8.1f us/step' % (k, n, t / args.steps * 1e6))
        if args.cprofile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            torch.cuda.synchronize()
            pr.enable()
            for _ in range(args.steps):
                step()
            pr.disable()
            torch.cuda.synchronize()
            st = pstats.Stats(pr, stream=sys.stdout)
            st.sort_stats('tottime').print_stats(18)
    zf._ZFUnetPlan.use_cplan = True


if __name__ == '__main__':
    main()
