"""debug: per-parameter gradient error of the odd-filter f32 ZF_UNET config on the GPU vs the fp64 oracle."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, 'segmentation-networks-benchmark_amd')]
import torch
from oracle import train_step_ref, zf_unet_ref
from lib.models.zf_unet import ZF_UNET
from lib.losses import BCEAndDiceLoss

B, S, F = 2, 64, int(os.environ.get('F', 6))
x, y = train_step_ref.synthetic_batch(B, S, seed=5)
sd = zf_unet_ref.default_init_state(filters=F, seed=2)
drop = zf_unet_ref.make_dropout_tables(F, B, 0.2, torch.Generator().manual_seed(3))
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
d64 = {k: v.double() for k, v in drop.items()}
_, lo64, g64 = train_step_ref.loss_and_grads(sd64, x.double(), y, 'bce_dice', drop=d64)
mode = os.environ.get('DBG', '')
if 'shift' in mode:
    keep = torch.zeros(12345, device='cuda')
if 'nan' in mode:
    junk = torch.full((1 << 28,), float('nan'), device='cuda')
    del junk
for rep in range(2):
    m = ZF_UNET(dropout_val=0.2, filters=F)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m.set_compute_dtype('f32').cuda().train()
    m.dropout_override = drop
    out = m(x.cuda())
    loss = BCEAndDiceLoss()(out, y.cuda())
    (B * loss).backward()
    torch.cuda.synchronize()
    print('rep', rep, 'logit err', float((out.detach().cpu().double() - lo64).abs().max()))
    for n, p in m.named_parameters():
        ref = g64[n]
        s = float(ref.abs().max())
        e = float((p.grad.cpu().double() - ref).abs().max())
        if s > 1e-9 and e / s > 1e-4:
            print('  %-40s scale %.3e relerr %.3e' % (n, s, e / s))
