"""debug: per-parameter gradient error (vs fp64 oracle) of the executor-driven models on the GPU."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, 'segmentation-networks-benchmark_amd'), os.path.join(root, 'tests')]
import numpy as np
import torch
import model_checks as mc
from lib.losses import BCEWithLogitsLossAndSmoothJaccard

which = sys.argv[1]
if which == 'tiramisu':
    g = np.load(os.path.join(root, 'tests/golden/tiramisu_small.npz'))
    model, fwd, x, y = mc.make_tiramisu(g)
elif which == 'linknet':
    model, fwd, x, y = mc.make_linknet(128)
else:
    model, fwd, x, y = mc.make_unet16()
B = x.shape[0]
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
pnames = set(n for n, _ in model.named_parameters())
lo32, loss32, g32 = mc._oracle_grads(fwd, sd, pnames, x, y, torch.float32, B)
lo64, loss64, g64 = mc._oracle_grads(fwd, sd, pnames, x, y, torch.float64, B)
model.set_compute_dtype('f32')
model.to('cuda').train()
for rep in range(2):
    model.zero_grad()
    out = model(x.cuda())
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.cuda())
    (B * loss).backward()
    torch.cuda.synchronize()
    print('rep %d logits err %.3e (oracle32 %.3e) loss err %.3e' % (rep, float((out.detach().cpu().double() - lo64).abs().max()),
          float((lo32.double() - lo64).abs().max()), abs(loss.item() - loss64)))
    rows = []
    for n, p in model.named_parameters():
        s = float(g64[n].abs().max())
        if s < 1e-12:
            continue
        rows.append((float((p.grad.cpu().double() - g64[n]).abs().max()) / s, float((g32[n] - g64[n]).abs().max()) / s, n))
    rows.sort(reverse=True)
    for e, er, n in rows[:8]:
        print('   %-45s e_prod %.3e e_ref %.3e' % (n, e, er))
    print('   median e_prod %.3e' % np.median([r[0] for r in rows]))
