#!/usr/bin/env python3
"""Per-layer timing of the BatchNorm / activation passes on the ZF_UNET bs=32 224x224 shapes (kernel tuning aid).
Each line: shape, then per kernel: time per launch and algorithmic GB/s (tensor passes x bytes)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd'))
sys.path.insert(0, ROOT)
import torch

from segnb import _native as nv
from segnb.engine import Runtime, View


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    rt = Runtime('cuda', 'bf16')
    N = args.batch
    tot = {}
    for lvl, (hw, C) in enumerate([(224, 32), (112, 64), (56, 128), (28, 256), (14, 512), (7, 1024)]):
        y = View.alloc(rt, N, hw, hw, C); y.t.normal_()
        out = View.alloc(rt, N, hw, hw, C)
        pool = View.alloc(rt, N, hw // 2, hw // 2, C)
        up = View.alloc(rt, N, 2 * hw, 2 * hw, C) if hw <= 112 else None
        g = View.alloc(rt, N, hw, hw, C); g.t.normal_()
        gp = View.alloc(rt, N, hw // 2, hw // 2, C); gp.t.normal_()
        gu = View.alloc(rt, N, 2 * hw, 2 * hw, C) if hw <= 112 else None
        dz = View.alloc(rt, N, hw, hw, C)
        coef = rt.zeros((4, C), torch.float32); coef[0] = 1.0; coef[3] = 1.0
        bcoef = rt.zeros((3, C), torch.float32); bcoef[0] = 1.0
        sums = rt.zeros((16, 2, C), torch.float64)
        drop = torch.ones(N, C, device='cuda')
        B = N * hw * hw * C * 2 / 1e3        # KB per tensor pass -> us * GB/s

        def fwd(o=None, p=None, u=None, d=None):
            return lambda: nv.call('segnb_bn_act_fwd', rt.code, y.ptr, y.ld, N, hw, hw, C, nv.ptr(coef), nv.ACT_RELU, 0.0,
                                   nv.ptr(d), None if o is None else o.ptr, 0 if o is None else o.ld,
                                   None if p is None else p.ptr, 0 if p is None else p.ld,
                                   None if u is None else u.ptr, 0 if u is None else u.ld, None, 0, rt.stream)

        def red(gd=None, gpp=None, guu=None, d=None):
            return lambda: nv.call('segnb_bn_act_bwd_reduce', rt.code, y.ptr, y.ld, N, hw, hw, C, nv.ptr(coef), nv.ACT_RELU,
                                   0.0, nv.ptr(d), None if gd is None else gd.ptr, 0 if gd is None else gd.ld,
                                   None if gpp is None else gpp.ptr, 0 if gpp is None else gpp.ld,
                                   None if guu is None else guu.ptr, 0 if guu is None else guu.ld, dz.ptr, dz.ld,
                                   nv.ptr(sums), None, 0, rt.stream)

        apply_ = lambda: nv.call('segnb_bn_bwd_apply', rt.code, y.ptr, y.ld, N, hw, hw, C, nv.ptr(coef), nv.ptr(bcoef),
                                 dz.ptr, dz.ld, dz.ptr, dz.ld, None, C, rt.stream)
        # the fused entry points the training step uses (finalize inside the pass): statistics of N(0, 1) data
        stats = rt.zeros((16, 2, C), torch.float64)
        stats[0, 1] = float(N * hw * hw)
        gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
        rm, rv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(1, dtype=torch.long, device='cuda')
        dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        fcoef = rt.zeros((4, C), torch.float32)
        fbcoef = rt.zeros((3, C), torch.float32)
        stats2 = rt.zeros((16, 2, C), torch.float64)
        fwd_fused = lambda: nv.call('segnb_bn_fwd_fused', rt.code, y.ptr, y.ld, N, hw, hw, C, C, nv.ptr(stats), nv.ptr(gamma),
                                    nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), nv.ptr(fcoef), None,
                                    nv.ACT_RELU, 0.0, None, out.ptr, out.ld, None, 0, None, 0, None, 0, rt.stream)
        apply_fused = lambda: nv.call('segnb_bn_bwd_apply_fused', rt.code, y.ptr, y.ld, N, hw, hw, C, C, nv.ptr(coef),
                                      nv.ptr(sums), nv.ptr(gamma), nv.ptr(fbcoef), nv.ptr(dgam), nv.ptr(dbet), 1,
                                      nv.ptr(stats2), dz.ptr, dz.ld, dz.ptr, dz.ld, rt.stream)
        red_sums = lambda: nv.call('segnb_bn_act_bwd_reduce', rt.code, y.ptr, y.ld, N, hw, hw, C, nv.ptr(coef), nv.ACT_RELU,
                                   0.0, None, g.ptr, g.ld, None, 0, None, 0, None, 0, nv.ptr(sums), None, 0, rt.stream)
        apply_fd = lambda: nv.call('segnb_bn_bwd_apply_fused_direct', rt.code, y.ptr, y.ld, N, hw, hw, C, C, nv.ptr(coef),
                                   nv.ptr(sums), nv.ptr(gamma), nv.ptr(fbcoef), nv.ptr(dgam), nv.ptr(dbet), 1, nv.ptr(stats2),
                                   nv.ACT_RELU, 0.0, g.ptr, g.ld, dz.ptr, dz.ld, rt.stream)

        def two_pass():
            red_sums()
            apply_fd()
        cases = [('fwd', fwd(o=out), 2.0), ('fwd(fused)', fwd_fused, 2.0), ('apply(fused)', apply_fused, 3.0), ('fwd+pool+drop', fwd(o=out, p=pool, d=drop), 2.25),
                 ('red(d)', red(gd=g), 3.0), ('red(d+pool)', red(gd=g, gpp=gp, d=drop), 3.25), ('apply', apply_, 3.0)]
        cases += [('red(pool)', red(gpp=gp), 2.25), ('red(d+pool,nodrop)', red(gd=g, gpp=gp), 3.25)]
        cases += [('red+apply(direct)', two_pass, 5.0)]
        if up is not None:
            cases += [('fwd(up)', fwd(u=up, d=drop), 5.0), ('red(up)', red(guu=gu, d=drop), 6.0)]
        line = '%3dx%-3d C=%-4d' % (hw, hw, C)
        for name, fn, passes in cases:
            us = timeit(fn, args.reps)
            tot[name] = tot.get(name, 0.0) + us
            line += '  %s %6.1f us %5.0f GB/s' % (name, us, passes * B / us)
        print(line)
    print('totals (us): ' + '  '.join('%s %.0f' % kv for kv in tot.items()))


if __name__ == '__main__':
    main()
