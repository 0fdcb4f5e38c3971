"""debug: diff every backward buffer between repetitions of the same step (finds the first racy kernel)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, 'segmentation-networks-benchmark_amd')]
import torch
from oracle import train_step_ref, zf_unet_ref
from lib.models.zf_unet import ZF_UNET, ENCODER, DECODER
from lib.losses import BCEAndDiceLoss

B, S, F = 2, 64, 6
x, y = train_step_ref.synthetic_batch(B, S, seed=5)
sd = zf_unet_ref.default_init_state(filters=F, seed=2)
drop = zf_unet_ref.make_dropout_tables(F, B, 0.2, torch.Generator().manual_seed(3))
m = ZF_UNET(dropout_val=0.2, filters=F)
m.load_state_dict({k: v.clone() for k, v in sd.items()})
m.set_compute_dtype('f32').cuda().train()
m.dropout_override = drop
xs, ys = x.cuda(), y.cuda()
snaps = []
order = ['df0']
for name, lvl in zip(reversed(DECODER), (0, 1, 2, 3, 4)):
    order += ['%s.l2.dz' % name, '%s.l2.bcoef' % name, 'db1_%d' % lvl, '%s.l1.dz' % name, '%s.l1.bcoef' % name, 'dcat_%d' % lvl]
for i in (5, 4, 3, 2, 1, 0):
    order += ['%s.l2.dz' % ENCODER[i], '%s.l2.bcoef' % ENCODER[i], 'da1_%d' % i, '%s.l1.dz' % ENCODER[i], '%s.l1.bcoef' % ENCODER[i]] + (['dp_%d' % i] if i > 0 else [])
for rep in range(int(os.environ.get('REPS', 8))):
    # restore BN running stats so every repetition is the same computation
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(sd[k])
    m.zero_grad()
    out = m(xs)
    loss = BCEAndDiceLoss()(out, ys)
    (B * loss).backward()
    torch.cuda.synchronize()
    eng = m._engine
    bufs = eng.buffers(B, S, S)
    snap = {}
    for k in order:
        if k in bufs:
            snap[k] = bufs[k].dense().clone()
        else:
            name, l, what = k.split('.')
            st = eng.stages[name][0 if l == 'l1' else 1]
            snap[k] = st.bcoef.clone() if what == 'bcoef' else list(st._bufs.values())[0]['dz'].dense().clone()
    snap['grad'] = eng.flat.flat_g.clone()
    snaps.append(snap)
ref = snaps[-1]
for r, sn in enumerate(snaps[:-1]):
    bad = []
    for k in order + ['grad']:
        d = float((sn[k].double() - ref[k].double()).abs().max())
        s = float(ref[k].abs().max()) + 1e-30
        if d / s > 1e-4:
            bad.append('%s %.2e' % (k, d / s))
    print('rep', r, 'vs last:', 'identical' if not bad else ' | '.join(bad[:6]))
