import os
"""debug tool: run one training step of a model on the ABI emulator (CPU) and on the GPU, checksumming every
tensor argument after every ABI call; report the first calls whose results differ."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, 'segmentation-networks-benchmark_amd'), os.path.join(root, 'tests')]
import bisect
import copy
import numpy as np
import torch
import model_checks as mc
from oracle import abi_emulator
from segnb import _native as nv
import segnb.engine as E
from lib.losses import BCEWithLogitsLossAndSmoothJaccard

which, dtype = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'f32')
if which == 'tiramisu':
    g = np.load(os.path.join(root, 'tests/golden/tiramisu_small.npz'))
    model, fwd, x, y = mc.make_tiramisu(g)
elif which == 'linknet':
    model, fwd, x, y = mc.make_linknet(128)
else:
    model, fwd, x, y = mc.make_unet16()
B = x.shape[0]

FULL = os.environ.get('FULL', '1') == '1'
registry = {}          # base data_ptr -> tensor


def reg(t):
    if t is not None and torch.is_tensor(t):
        st = t.untyped_storage()
        registry[st.data_ptr()] = (st.nbytes(), t)


orig_ptr = nv.ptr


def ptr(t, offset_elems=0):
    reg(t)
    return orig_ptr(t, offset_elems)


nv.ptr = ptr
E.nv.ptr = ptr
orig_view_ptr = E.View.ptr.fget
E.View.ptr = property(lambda self: (reg(self.t), orig_view_ptr(self))[1])
orig_call = nv.call
log = []


def traced(name, *args):
    orig_call(name, *args)
    starts = sorted(registry)
    rec = []
    offs = {}
    for pos, a in enumerate(args):
        if isinstance(a, int) and a > (1 << 24):
            i = bisect.bisect_right(starts, a) - 1
            if i >= 0 and a < starts[i] + registry[starts[i]][0]:
                t = registry[starts[i]][1]
                if t.is_floating_point():
                    st = t.untyped_storage()
                    full = torch.empty(0, dtype=t.dtype, device=t.device).set_(st)
                    d = full.double()
                    offs[pos] = (a - starts[i]) // t.element_size()
                    rec.append((pos, d.cpu().clone() if FULL else torch.stack([d.sum(), d.abs().sum()]).cpu()))
    log.append((name, rec, [a if isinstance(a, (int, float)) or a is None else '?' for a in args], offs))


nv.call = traced
E.nv.call = traced


def run(device):
    global log
    log = []
    registry.clear()
    m = copy.deepcopy(model)
    m.set_compute_dtype(dtype)
    m.to(device).train()
    out = m(x.to(device))
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.to(device))
    (B * loss).backward()
    if device != 'cpu':
        torch.cuda.synchronize()
    return log


os.environ['SEGNB_TEST_HARNESS'] = '1'       # a tool, not the product: allowed to install the emulator
nv.set_backend_for_testing(abi_emulator.AbiEmulator())
ref = run('cpu')
nv.set_backend_for_testing(None)
got = run('cuda')
print('calls: emulator %d, gpu %d' % (len(ref), len(got)))
tol = 1e-5 if dtype == 'f32' else 3e-3
shown = 0
detail = int(os.environ.get('DETAIL', -1))
for i, ((n1, r1, a1, o1), (n2, r2, a2, o2)) in enumerate(zip(ref, got)):
    assert n1 == n2, (i, n1, n2)
    for (p1, c1), (p2, c2) in zip(r1, r2):
        if FULL:
            if n1 in ('segnb_bn_stats', 'segnb_bn_act_bwd_reduce') and c1.numel() % 32 == 0 and c1.dtype == torch.float64 \
                    and p1 in (7, 19):
                c1, c2 = c1.view(16, -1).sum(0), c2.view(16, -1).sum(0)      # replicated accumulators
            scale = float(c1.abs().max()) + 1e-30
            err = float((c1 - c2).abs().max()) / scale
            rms = float((c1 - c2).pow(2).mean().sqrt() / (c1.pow(2).mean().sqrt() + 1e-30))
            if err > tol * 10:
                print('#%d %s arg %d: max err %.3e of max, rel rms %.3e (n=%d)' % (i, n1, p1, err, rms, c1.numel()))
                shown += 1
        else:
            d = abs(float(c1[0] - c2[0]))
            if d > tol * float(c1[1]) + 1e-30:
                print('#%d %s arg %d: emulator sum %.6e abs %.6e | gpu sum %.6e abs %.6e' % (
                    i, n1, p1, float(c1[0]), float(c1[1]), float(c2[0]), float(c2[1])))
                shown += 1
    if i == detail:
        N, H, W, Cp = a1[3:7]
        npix = N * H * W
        T1, T2 = dict(r1), dict(r2)
        def view(T, o, pos, ld, C=Cp):
            return T[pos].reshape(-1)[o[pos]:].as_strided((npix, C), (ld, 1))
        y1, y2 = view(T1, o1, 1, a1[2]), view(T2, o2, 1, a2[2])
        co1, co2 = T1[7].view(4, Cp), T2[7].view(4, Cp)
        g1, g2 = view(T1, o1, 11, a1[12]), view(T2, o2, 11, a2[12])
        dz1, dz2 = view(T1, o1, 17, a1[18]), view(T2, o2, 17, a2[18])
        print('act', a1[8], 'slope', a1[9], 'has drop', a1[10] is not None, 'res', a1[20] is not None, 'N H W Cp', N, H, W, Cp)
        print('y equal:', bool((y1 == y2).all()), 'max dy', float((y1 - y2).abs().max()), ' coef max d', float((co1 - co2).abs().max()),
              ' g max d', float((g1 - g2).abs().max()))
        z1 = (y1 - co1[2]) * co1[0] + co1[1]
        z2 = (y2 - co2[2]) * co2[0] + co2[1]
        bad = ((dz1 - dz2).abs() > 1e-3 * float(dz1.abs().max())).nonzero()
        print('bad elements', len(bad), 'of', dz1.numel(), ' channels:', sorted(set(bad[:, 1].tolist()))[:40])
        for q in bad[:12].tolist():
            pp, c = q
            print('   pix %d c %d: dz emu %.4e gpu %.4e g %.4e  z emu %.4e gpu %.4e y %.6e mean %.6e scale %.4e' % (
                pp, c, float(dz1[pp, c]), float(dz2[pp, c]), float(g1[pp, c]), float(z1[pp, c]), float(z2[pp, c]),
                float(y1[pp, c]), float(co1[2, c]), float(co1[0, c])))
    if shown >= 16 and i >= detail:
        break
print('done, %d mismatching (call, arg) pairs shown' % shown)
