import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [root, root + '/segmentation-networks-benchmark_amd']
import torch
from lib.models.zf_unet import ZF_UNET
from lib.losses import BCEAndDiceLoss
from segnb import optim
m = ZF_UNET().cuda().train()
opt = optim.SGD(m.parameters(), lr=1e-3)
crit = BCEAndDiceLoss()
x = torch.randn(32, 3, 224, 224).cuda(); y = (torch.rand(32, 1, 224, 224) > 0.7).long().cuda()
def step():
    opt.zero_grad(); out = m(x); loss = crit(out, y); (x.size(0) * loss).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
