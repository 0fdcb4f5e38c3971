"""debug: elementwise diff of the first BN-backward reduce between repetitions."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, 'segmentation-networks-benchmark_amd')]
import torch
from oracle import train_step_ref, zf_unet_ref
from lib.models.zf_unet import ZF_UNET
from lib.losses import BCEAndDiceLoss
from segnb import _native as nv
import segnb.engine as E

B, S, F = 2, int(os.environ.get("S", 64)), 6
x, y = train_step_ref.synthetic_batch(B, S, seed=5)
sd = zf_unet_ref.default_init_state(filters=F, seed=2)
drop = zf_unet_ref.make_dropout_tables(F, B, 0.2, torch.Generator().manual_seed(3))
m = ZF_UNET(dropout_val=0.2, filters=F)
m.load_state_dict({k: v.clone() for k, v in sd.items()})
m.set_compute_dtype('f32').cuda().train()
m.dropout_override = drop
xs, ys = x.cuda(), y.cuda()


def step():
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(sd[k])
    m.zero_grad()
    out = m(xs)
    loss = BCEAndDiceLoss()(out, ys)
    (B * loss).backward()
    return out


step()
torch.cuda.synchronize()
eng = m._engine
bufs = eng.buffers(B, S, S)
st = eng.stages['up_conv_224'][1]
sb = list(st._bufs.values())[0]
snap = None
orig = nv.call
count = [0]


def traced(name, *args):
    orig(name, *args)
    if snap is not None and name == 'segnb_bn_act_bwd_reduce':
        count[0] += 1
        if count[0] == 1:
            snap.update(y=sb['y'].dense().clone(), dz=sb['dz'].dense().clone(), df0=bufs['df0'].dense().clone(),
                        coef=st.coef.clone(), drop=bufs['drop']['up_conv_224'].clone(), sums=st.sums.clone())


nv.call = traced
snaps = []
for rep in range(6):
    snap = {}
    count[0] = 0
    out = step()
    torch.cuda.synchronize()
    snap['logits'] = out.detach().clone()
    snaps.append(snap)
    snap = None
ref = snaps[-1]
for r, sn in enumerate(snaps[:-1]):
    line = []
    for k in ('logits', 'y', 'coef', 'drop', 'df0', 'dz', 'sums'):
        ne = (sn[k] != ref[k])
        line.append('%s:%d/%d(max %.2e)' % (k, int(ne.sum()), ne.numel(), float((sn[k].double() - ref[k].double()).abs().max())))
    print('rep', r, ' '.join(line))
    ne = (sn['dz'] != ref['dz']).nonzero()
    for idx in ne[:5].tolist():
        n, h, w, c = idx
        yv = float(ref['y'][n, h, w, c]); co = ref['coef']
        z = (yv - float(co[2, c])) * float(co[0, c]) + float(co[1, c])
        print('     at', idx, 'dz %.4e vs %.4e  y %.6e z %.4e g %.4e drop %.2f' % (
            float(sn['dz'][n, h, w, c]), float(ref['dz'][n, h, w, c]), yv, z, float(ref['df0'][n, h, w, c]),
            float(ref['drop'][n, c])))
