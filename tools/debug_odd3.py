"""debug: checksum every buffer an ABI call touches, right after the call (stream-ordered, no host sync); diff
the checksum sequences of repeated identical steps to find the first call whose result is not reproducible."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, 'segmentation-networks-benchmark_amd')]
import bisect
import torch
from oracle import train_step_ref, zf_unet_ref
from lib.models.zf_unet import ZF_UNET, ENCODER, DECODER
from lib.losses import BCEAndDiceLoss
from segnb import _native as nv
import segnb.engine as E

B, S, F = 2, 64, 6
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


step()
torch.cuda.synchronize()
eng = m._engine
tracked = {}
for k, v in eng.buffers(B, S, S).items():
    if isinstance(v, E.View):
        tracked[k] = v.t
    elif torch.is_tensor(v):
        tracked[k] = v
tracked['drop_flat'] = eng.buffers(B, S, S)['drop_flat']
tracked['flat_g'] = eng.flat.flat_g
tracked['flat_p'] = eng.flat.flat_p
for name, sts in eng.stages.items():
    for l, st in zip(('l1', 'l2'), sts):
        for kk, bb in list(st._bufs.values())[0].items():
            tracked['%s.%s.%s' % (name, l, kk)] = bb.t
        for attr in ('stats', 'sums', 'coef', 'bcoef'):
            tracked['%s.%s.%s' % (name, l, attr)] = getattr(st, attr)
        for attr, val in vars(st.conv).items():
            if torch.is_tensor(val) and val.is_cuda and val.is_floating_point():
                tracked['%s.%s.conv.%s' % (name, l, attr)] = val
            if isinstance(val, dict):
                for k2, v2 in val.items():
                    if torch.is_tensor(v2) and v2.is_cuda and v2.is_floating_point():
                        tracked['%s.%s.conv.%s[%s]' % (name, l, attr, k2)] = v2
ranges = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), n) for n, t in tracked.items())
starts = [r[0] for r in ranges]
print('tracking %d buffers' % len(ranges))
log = None
orig_call = nv.call


def traced(name, *args):
    orig_call(name, *args)
    if log is None:
        return
    names = set()
    for a in args:
        if isinstance(a, int) and a > (1 << 32):
            i = bisect.bisect_right(starts, a) - 1
            if i >= 0 and ranges[i][0] <= a < ranges[i][1]:
                names.add(ranges[i][2])
    rec = {}
    for n in names:
        t = tracked[n].double()
        rec[n] = torch.stack([t.sum(), t.abs().sum()])
    log.append((name, rec))


nv.call = traced
import segnb.engine as E
logs = []
for rep in range(int(os.environ.get('REPS', 6))):
    log = []
    step()
    torch.cuda.synchronize()
    logs.append([(n, {k: v.cpu() for k, v in r.items()}) for n, r in log])
    log = None
ref = logs[-1]
for r, lg in enumerate(logs[:-1]):
    assert len(lg) == len(ref)
    msgs = []
    for i, ((n1, r1), (n2, r2)) in enumerate(zip(lg, ref)):
        for k in r1:
            a, b = r1[k], r2[k]
            d = float((a - b).abs().max())
            if d > 1e-6 * float(b[1]) + 1e-30:
                msgs.append('#%d %s %s sum %.6e vs %.6e (abs %.3e)' % (i, n1, k, float(a[0]), float(b[0]), float(b[1])))
        if len(msgs) >= 6:
            break
    print('rep %d:' % r, 'reproducible' if not msgs else '\n   ' + '\n   '.join(msgs[:6]))
