"""The activation-mask data gradient (segnb_conv_fprop_bnreduce with coef NULL on conv_fprop_ws_kernel's MASK instantiation) against
the plain data gradient + the producing layer's own mask pass (segnb_bn_act_bwd_reduce with coef NULL), launches alone, HIP events:
the conv + ReLU stacks of lib/models/unet16.py:73-102 at the row's configuration (bs=4, 1024 x 1024).

    python tools/mask_bench.py [--reps 10] [--batch 4] [--size 1024]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'segmentation-networks-benchmark_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

from segnb import _native as nv
from segnb.engine import ConvOp, Runtime, View

# (producer -> consumer) pairs whose producer's mask pass the consumer's data gradient absorbs: level divisor, C(producer out), C(consumer out)
LAYERS = [('enc0.0 <- enc0.1', 1, 64, 64), ('enc1.0 <- enc1.1', 2, 128, 128), ('enc2.0 <- enc2.1', 4, 256, 256),
          ('enc3.0 <- enc3.1', 8, 512, 512), ('enc4.0 <- enc4.1', 16, 512, 512)]


def timed(fn, reps):
    for _ in range(3):
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
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=1024)
    args = ap.parse_args()
    nv.load()
    rt = Runtime('cuda', 'bf16')
    N = args.batch
    tot = [0.0, 0.0, 0.0]
    for name, div, C1, C2 in LAYERS:
        S = args.size // div
        w = torch.randn(C2, C1, 3, 3, device='cuda') * (2.0 / (C1 * 9)) ** 0.5
        op = ConvOp(rt, w, None, [(C1, C1)], 1, 1, False, True)
        op.pack(S, S)
        dy = View.alloc(rt, N, S, S, C2)
        dy.t.normal_()
        a1 = View.alloc(rt, N, S, S, C1)
        a1.t.normal_().clamp_(min=0)
        dx, dz = View.alloc(rt, N, S, S, C1), View.alloc(rt, N, S, S, C1)
        sums = rt.zeros((16, 2, C1), torch.float64)
        st = rt.stream
        ok = op.dgrad_actmask_ok(dy, dx)
        t_plain = timed(lambda: op.dgrad(dy, dx), args.reps)
        t_pass = timed(lambda: nv.call('segnb_bn_act_bwd_reduce', rt.code, a1.ptr, a1.ld, N, S, S, C1, None, nv.ACT_RELU, 0.0, None,
                                       dx.ptr, dx.ld, None, 0, None, 0, dz.ptr, dz.ld, nv.ptr(sums), None, 0, st), args.reps)
        t_mask = timed(lambda: op.dgrad(dy, dz, bn_reduce=(a1, None, sums, nv.ACT_RELU, 0.0)), args.reps) if ok else float('nan')
        gb = 3.0 * N * S * S * C1 * 2 / 1e9
        print('%-18s %4d^2 %3d<-%3d  dgrad %7.1f us   mask pass %7.1f us (%.2f TB/s)   dgrad+mask fused %7.1f us   saved %7.1f us'
              % (name, S, C1, C2, t_plain, t_pass, gb / t_pass * 1e3, t_mask, t_plain + t_pass - t_mask))
        tot[0] += t_plain
        tot[1] += t_pass
        tot[2] += t_mask
    print('totals (us): dgrad %.0f  mask pass %.0f  fused %.0f' % tuple(tot))


if __name__ == '__main__':
    main()
