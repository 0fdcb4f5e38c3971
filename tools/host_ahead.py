"""Is the host ahead of the GPU in the bench loop?  Per step: host time when the step's launches have been issued vs the
GPU time its last kernel ended (events), over a free-running loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'segmentation-networks-benchmark_amd')); sys.path.insert(0, ROOT)
import torch
from lib import losses as L
from lib.models.zf_unet import ZF_UNET
from segnb import optim
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = ZF_UNET().to(dev).train()
crit = L.BCEAndDiceLoss()
opt = optim.SGD(model.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1234)
x = torch.randn(32, 3, 224, 224, generator=g).to(dev)
y = (torch.rand(32, 1, 224, 224, generator=g) > 0.7).long().to(dev)
def step():
    opt.zero_grad(); out = model(x); loss = crit(out, y); (32 * loss).backward(); opt.step()
for _ in range(6): step()
torch.cuda.synchronize()
N = 12
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
evs[0].record(); t0 = time.perf_counter()
for i in range(N):
    ta = time.perf_counter()
    opt.zero_grad(); tb = time.perf_counter()
    out = model(x); tc = time.perf_counter()
    loss = crit(out, y); td = time.perf_counter()
    (32 * loss).backward(); te = time.perf_counter()
    opt.step(); tf = time.perf_counter()
    evs[i + 1].record()
    host.append((ta - t0, tb - ta, tc - tb, td - tc, te - td, tf - te))
torch.cuda.synchronize()
for i in range(N):
    print('step %2d: host issued at %7.2f ms (zero_grad %.2f fwd %.2f loss %.2f bwd %.2f opt %.2f)  gpu done at %7.2f ms'
          % ((i, host[i][0] * 1e3 + sum(host[i][1:]) * 1e3) + tuple(v * 1e3 for v in host[i][1:]) + (evs[0].elapsed_time(evs[i + 1]),)))
