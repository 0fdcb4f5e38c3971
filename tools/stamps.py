#!/usr/bin/env python3
"""In-kernel timeline of one block of conv_fprop_ws_kernel (timing build: segnb_tune fprop_dma_dbg=32).

    python tools/stamps.py --layer enc3.l2 [--what fprop]
Per tap: matrix wave 0 (start .. MFMAs issued .. stores issued .. past barrier), weight wave, halo wave; cycles."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'segmentation-networks-benchmark_amd')]
import torch
from segnb import _native as nv
from segnb import convplan as cp
from segnb.engine import ConvOp, Runtime, View

ap = argparse.ArgumentParser()
ap.add_argument('--hw', type=int, default=28)
ap.add_argument('--ci', type=int, default=256)
ap.add_argument('--co', type=int, default=256)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--steps', type=int, default=45)
ap.add_argument('--extra', type=int, default=0, help='extra fprop_dma_dbg bits (1 no weights, 2 no halo, 16 no fragment reads)')
ap.add_argument('--light', action='store_true', help='one stamp per tap (matrix wave, tap start) only: 64 instead of 32')
args = ap.parse_args()
rt = Runtime('cuda', 'bf16')
wt = torch.randn(args.co, args.ci, 3, 3, device='cuda') * 0.05
op = ConvOp(rt, wt, torch.zeros(args.co, device='cuda'), [(args.ci, args.ci)], 1, 1, False, True)
op.pack(args.hw, args.hw)
xv = View.alloc(rt, args.batch, args.hw, args.hw, op.Cip); xv.t.normal_()
yv = View.alloc(rt, args.batch, args.hw, args.hw, op.Cop)
stats = rt.zeros((16, 2, op.Cop), torch.float64)
for _ in range(3):
    op.fprop(xv, yv, stats)
torch.cuda.synchronize()
nv.call('segnb_tune', b'fprop_dma_dbg', (64 if args.light else 32) | args.extra)
op.fprop(xv, yv, stats)
torch.cuda.synchronize()
nv.call('segnb_tune', b'fprop_dma_dbg', 0)
buf = (ctypes.c_ulonglong * (3 * 256 * 4))()
nv.call('segnb_debug_stamps', ctypes.cast(buf, ctypes.c_void_p))
import numpy as np
st = np.array(buf[:], dtype=np.uint64).reshape(3, 256, 4).astype(np.int64)
t0 = st[0, 0, 0]
if args.light:
    d = np.diff(st[0, :args.steps + 1, 0])
    print('tap lengths (cycles, matrix wave 0, one stamp per tap):', d.tolist())
    print('median %d  mean %.0f' % (np.median(d), d.mean()))
    sys.exit(0)
print('tap | matrix: start  +mfma_issued +stores +barrier | weight: start +issued +landed +barrier | halo: start +issued +landed +barrier | tap length')
for sidx in range(args.steps):
    m, w, h = st[0, sidx], st[1, sidx], st[2, sidx]
    nxt = st[0, sidx + 1, 0] - m[0]
    print('%3d | %7d %5d %5d %5d | %7d %5d %5d %5d | %7d %5d %5d %5d | %6d' % (
        sidx, m[0] - t0, m[1] - m[0], m[2] - m[1], m[3] - m[2], w[0] - t0, w[1] - w[0], w[2] - w[1], w[3] - w[2],
        h[0] - t0, h[1] - h[0], h[2] - h[1], h[3] - h[2], nxt))
