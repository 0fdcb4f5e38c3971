"""can the define-by-run executor be captured into a HIP graph?  (host enqueue vs total, eager vs replay)"""
import sys, os, time, warnings
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [root, root + '/segmentation-networks-benchmark_amd']
import torch
from lib.losses import BCEWithLogitsLossAndSmoothJaccard
from lib.models.tiramisu import FCDenseNet103
from lib.models.linknet import LinkNet34
from segnb import optim

for name, ctor, B, S in (('FCDenseNet103', lambda: FCDenseNet103(n_classes=1), 8, 256), ('LinkNet34', LinkNet34, 16, 512)):
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ctor().cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    x = torch.randn(B, 3, S, S).cuda(); y = (torch.rand(B, 1, S, S) > 0.7).long().cuda()
    crit = BCEWithLogitsLossAndSmoothJaccard(); opt = optim.SGD(m.parameters(), lr=1e-3)
    def step():
        opt.zero_grad(set_to_none=False)
        loss = crit(m(x), y); (B * loss).backward(); opt.step(); return loss
    for _ in range(3): l = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): l = step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%s eager: host enqueue %.1f ms/step, total %.1f ms/step, loss %.5f' % (name, (t1 - t0) * 100, (t2 - t0) * 100, float(l)))
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sl = step()
        torch.cuda.synchronize()
        for _ in range(2): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print('%s graph replay: %.1f ms/step, loss %.5f' % (name, (time.perf_counter() - t0) * 100, float(sl)))
    except Exception as e:
        print(name, 'graph capture failed:', repr(e)[:300])
