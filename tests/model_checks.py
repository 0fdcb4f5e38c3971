"""Shared checks for the executor-driven models (UNet16, LinkNet34, FCDenseNet): product (on `device`, through
the C ABI) vs the oracle restatements.  The fp32 comparison is noise-aware: the oracle is run in fp64 too, and the
product's error is bounded by the fp32 oracle's own distance from fp64 (deep BN nets at random init amplify fp32
summation noise by orders of magnitude on the way back to the first layers)."""
import warnings

import numpy as np
import pytest
import torch

from oracle import linknet_ref, losses_ref, tiramisu_ref, unet16_ref

TIRAMISU_CFG = dict(in_channels=3, down_blocks=(2, 3), up_blocks=(3, 2), bottleneck_layers=2, growth_rate=8,
                    out_chans_first_conv=16, n_classes=1)


def _oracle_grads(forward, sd, pnames, x, y, dtype, B):
    leaves = {}
    for k, v in sd.items():
        if k in pnames:
            leaves[k] = v.clone().to(dtype).requires_grad_(True)
        else:
            leaves[k] = v.clone().to(dtype) if v.is_floating_point() else v.clone()
    logits = forward(leaves, x.to(dtype))
    loss = losses_ref.bce_jaccard(logits, y)
    (B * loss).backward()
    return logits.detach(), loss.item(), {k: leaves[k].grad.double() for k in pnames}


def check_against_oracle(model, forward, x, y, device, dtype='f32', min_cos=0.99999):
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    B = x.shape[0]
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    pnames = set(n for n, _ in model.named_parameters())
    lo32, loss32, g32 = _oracle_grads(forward, sd, pnames, x, y, torch.float32, B)
    lo64, loss64, g64 = _oracle_grads(forward, sd, pnames, x, y, torch.float64, B)
    model.set_compute_dtype(dtype)
    model.to(device).train()
    out = model(x.to(device))
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.to(device))
    (B * loss).backward()
    if device != 'cpu':
        torch.cuda.synchronize()
    scale = float(lo64.abs().max())
    lerr = float((out.detach().cpu().double() - lo64).abs().max())
    if dtype == 'f32':
        assert lerr <= 1e-4 * scale + 5 * float((lo32.double() - lo64).abs().max()), ('logits', lerr, scale)
        assert abs(loss.item() - loss64) < 1e-5
    else:
        assert abs(loss.item() - loss64) < 5e-3
    got = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}
    ga = torch.cat([got[n].reshape(-1) for n in sorted(pnames)])
    gr = torch.cat([g64[n].reshape(-1) for n in sorted(pnames)])
    cos = float((ga * gr).sum() / (ga.norm() * gr.norm()))
    if dtype == 'f32':
        for n in pnames:
            s = float(g64[n].abs().max())
            if s < 1e-7 * float(gr.abs().max()):
                continue                                     # analytically-zero gradients (conv bias under BN)
            # Elementwise gradient agreement is NOT asserted tightly here: one ReLU whose pre-activation is within
            # an fp32 ulp of zero flips between two correct implementations and moves every upstream gradient by
            # 1e-3..1e-2 in these small random-init nets (measured; see tests/abi_replay.py, which holds every
            # launch of the same plan to the per-op tolerance instead).  Bound: the relative L2 error per tensor.
            rel = float((got[n] - g64[n]).norm() / (g64[n].norm() + 1e-30))
            assert rel <= 0.1, (n, rel)
        assert cos > min_cos, cos
    else:
        assert np.isfinite(cos) and cos > 0.5, cos
    return cos


def make_unet16():
    from lib.models.unet16 import UNet16
    torch.manual_seed(0)
    m = UNet16(num_filters=8)
    g = torch.Generator().manual_seed(1)
    return m, unet16_ref.forward, torch.randn(2, 3, 32, 64, generator=g), (torch.rand(2, 1, 32, 64, generator=g) > 0.7).long()


def make_linknet(size=64):
    from lib.models.linknet import LinkNet34
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = LinkNet34()
    m.finaldrop1.p = 0.0
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, size, size + 32, generator=g)
    y = (torch.rand(2, 1, size, size + 32, generator=g) > 0.7).long()
    return m, (lambda sd, xx: linknet_ref.forward(sd, xx, True)), x, y


def make_tiramisu(golden):
    from lib.models.tiramisu import FCDenseNet
    m = FCDenseNet(**TIRAMISU_CFG)
    m.load_state_dict({k[3:]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith('sd/')})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    fwd = lambda sd, xx: tiramisu_ref.forward(sd, xx, TIRAMISU_CFG['down_blocks'], TIRAMISU_CFG['up_blocks'],
                                              TIRAMISU_CFG['bottleneck_layers'], True)
    return m, fwd, torch.from_numpy(golden['x']), torch.from_numpy(golden['y'])


def make_tiramisu57_golden(golden):
    """FCDenseNet57 (growth 12: padded slices) with the seeded fill the generator loaded into the reference's"""
    from lib.models.tiramisu import FCDenseNet57
    m = FCDenseNet57(n_classes=1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    return _load_seeded(m, golden)


def check_tiramisu_golden(model, golden, device, dtype='f32'):
    """FCDenseNet vs the fixture produced by the REFERENCE's lib/models/tiramisu.py."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    x, y = torch.from_numpy(golden['x']), torch.from_numpy(golden['y'])
    model.set_compute_dtype(dtype)
    model.to(device).train()
    out = model(x.to(device))
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.to(device))
    (2 * loss).backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), golden['train_logits'], rtol=1e-4, atol=1e-5)
    assert abs(loss.item() - float(golden['loss_bce_jaccard'])) < 1e-5
    gmax = max(np.abs(golden['grad/' + n]).max() for n, _ in model.named_parameters())
    for n, p in model.named_parameters():
        ref = golden['grad/' + n]
        if np.abs(ref).max() < 1e-6 * gmax:
            continue
        # 3e-2: a single flipped ReLU (|z| < 1e-6) moved gradients by up to 1.2e-2 on MI355X; see abi_replay.py
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-2 * np.abs(ref).max() + 1e-7, n
    for n, b in model.named_buffers():
        np.testing.assert_allclose(b.cpu().numpy(), golden['buf/' + n], rtol=1e-4, atol=1e-5, err_msg=n)
    model.eval()
    with torch.no_grad():
        ev = model(x.to(device))
    np.testing.assert_allclose(ev.cpu().numpy(), golden['eval_logits'], rtol=1e-3, atol=1e-4)


# ---- UNet16 / LinkNet34 vs the fixtures produced by the REFERENCE's own wiring code (make_golden.py gen_unet16 /
# gen_linknet: lib/models/unet16.py:52-131, lib/models/linknet.py:5-90 run on torch.nn stand-ins for torchvision /
# inplace_abn).  Weights: oracle.fill.seeded_state, loaded into the product module here exactly as the generator
# loaded them into the reference module.
def make_unet16_golden(golden):
    from lib.models.unet16 import UNet16
    m = UNet16(num_filters=8)
    return _load_seeded(m, golden), unet16_ref.forward


def make_unet16_bilinear_golden(golden):
    """UNet16 with every DecoderBlock on its bilinear branch (unet16.py:42-46; UNet16(..., is_deconv=False) here)"""
    from lib.models.unet16 import UNet16
    m = UNet16(num_filters=8, is_deconv=False)
    return _load_seeded(m, golden), (lambda sd, x: unet16_ref.forward(sd, x, is_deconv=False))


def make_linknet_golden(golden):
    from lib.models.linknet import LinkNet34
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = LinkNet34()
    m.finaldrop1.p = 0.0
    return _load_seeded(m, golden), None


def _load_seeded(m, golden):
    from oracle import fill
    sd = fill.seeded_state(m.state_dict(), int(golden['seed']))
    assert list(sd.keys()) == [str(k) for k in golden['sd_keys']], 'state_dict layout differs from the reference'
    assert [v.numel() for v in sd.values()] == [int(n) for n in golden['sd_numel']]
    m.load_state_dict(sd)
    return m


def check_oracle_golden(forward_train, forward_eval, model, golden):
    """The oracle restatement (functional, on the module's state_dict) vs the reference fixture: this is the PIN."""
    x, y = torch.from_numpy(golden['x']), torch.from_numpy(golden['y'])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    pnames = [n for n, _ in model.named_parameters()]
    with torch.no_grad():
        ev = forward_eval({k: v.clone() for k, v in sd.items()}, x)
    np.testing.assert_allclose(ev.numpy(), golden['eval_logits'], rtol=1e-4, atol=2e-5)
    leaves = {k: (v.clone().requires_grad_(True) if k in pnames else v.clone()) for k, v in sd.items()}
    # aliased keys (UNet16: encoder.N.* == convK.M.*) must be ONE leaf, or the gradient splits between the aliases
    seen = {}
    for n, p in model.state_dict(keep_vars=True).items():
        if id(p) in seen:
            leaves[n] = leaves[seen[id(p)]]
        else:
            seen[id(p)] = n
    logits = forward_train(leaves, x)
    np.testing.assert_allclose(logits.detach().numpy(), golden['train_logits'], rtol=1e-4, atol=2e-5)
    loss = losses_ref.bce_jaccard(logits, y)
    assert abs(loss.item() - float(golden['loss_bce_jaccard'])) < 1e-6
    (x.shape[0] * loss).backward()
    _check_grads_golden({n: leaves[n].grad for n in pnames}, golden, 1e-3, 2e-3)
    for k in golden.files:
        if k.startswith('buf/') and 'num_batches' not in k:
            np.testing.assert_allclose(leaves[k[4:]].detach().numpy(), golden[k], rtol=1e-4, atol=1e-6, err_msg=k)


def _check_grads_golden(grads, golden, norm_rtol, probe_rtol):
    names = [str(n) for n in golden['grad_names']]
    norms = dict(zip(names, golden['grad_norms']))
    gmax = max(norms.values())
    for n in names:
        g = grads[n].detach().cpu().double().reshape(-1)
        if norms[n] < 1e-6 * gmax:
            continue                                          # analytically zero (conv bias in front of BatchNorm)
        assert abs(float(g.norm()) - norms[n]) <= norm_rtol * norms[n], (n, float(g.norm()), norms[n])
        ref = golden['gval/' + n].astype(np.float64)
        got = g.numpy()[golden['gidx/' + n]]
        assert np.abs(got - ref).max() <= probe_rtol * max(np.abs(ref).max(), norms[n] / np.sqrt(g.numel())), n


def check_product_golden(model, golden, device, dtype='f32'):
    """The product module (through the C ABI on `device`) vs the reference fixture."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore
    x, y = torch.from_numpy(golden['x']), torch.from_numpy(golden['y'])
    B = x.shape[0]
    model.set_compute_dtype(dtype)
    model.to(device)
    model.eval()
    with torch.no_grad():
        ev = model(x.to(device))
    scale = float(np.abs(golden['eval_logits']).max())
    tol = 2e-4 if dtype == 'f32' else 6e-2
    assert float(np.abs(ev.cpu().numpy() - golden['eval_logits']).max()) <= tol * scale
    model.train()
    out = model(x.to(device))
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.to(device))
    iou = JaccardScore()(out, y.to(device))
    (B * loss).backward()
    scale = float(np.abs(golden['train_logits']).max())
    terr = float(np.abs(out.detach().cpu().numpy() - golden['train_logits']).max())
    print('%s train logits max|d| %.3e of scale %.3e' % (dtype, terr, scale))
    # training-mode forward of the bf16 path: every BatchNorm re-normalises with statistics of bf16-rounded tensors, so
    # the storage noise compounds over the 36 (LinkNet34) normalised layers -- bounded at 0.2 of the logit scale, with
    # the loss / IoU bounds below as the functional check
    assert terr <= (tol if dtype == 'f32' else 0.2) * scale, (terr, scale)
    dl, di = abs(loss.item() - float(golden['loss_bce_jaccard'])), abs(iou.item() - float(golden['iou']))
    if dtype == 'f32':
        assert dl < 1e-5 and di < 1e-4, (dl, di)              # north_star tolerances
        # per-tensor gradient norms within 1e-2, probed entries within 5e-2 of the tensor's scale: a ReLU whose
        # pre-activation is within an fp32 ulp of zero flips between two correct implementations (abi_replay.py)
        _check_grads_golden({n: p.grad for n, p in model.named_parameters()}, golden, 1e-2, 5e-2)
        for k in golden.files:
            if k.startswith('buf/') and 'num_batches' not in k:
                b = dict(model.named_buffers())[k[4:]]
                np.testing.assert_allclose(b.cpu().numpy(), golden[k], rtol=1e-4, atol=1e-5, err_msg=k)
    else:
        assert dl < 5e-3 and di < 5e-3, (dl, di)
        names = [str(n) for n in golden['grad_names']]
        gmax = float(golden['grad_norms'].max())
        for n, p in model.named_parameters():
            ref = float(golden['grad_norms'][names.index(n)])
            if ref > 1e-3 * gmax:
                assert abs(float(p.grad.norm()) - ref) < 0.25 * ref, (n, float(p.grad.norm()), ref)
    return dl, di


# ---- the rest of the loss constructor surface (lib/losses.py:47,84) vs values the REFERENCE produced (losses.npz)
EXTRA_LOSS_CASES = ['bce_sum', 'bce_none', 'focal_g15_mean', 'focal_g3_sum', 'focal_g0_sum']


def check_extra_loss_case(golden, tag, device):
    from lib import losses as L
    crit = {'bce_sum': lambda: L.BCEWithSigmoidLoss(size_average=False),
            'bce_none': lambda: L.BCEWithSigmoidLoss(reduce=False),
            'focal_g15_mean': lambda: L.FocalLossBinary(gamma=1.5),
            'focal_g3_sum': lambda: L.FocalLossBinary(gamma=3, size_average=False),
            'focal_g0_sum': lambda: L.FocalLossBinary(gamma=0, size_average=False)}[tag]()
    x = torch.from_numpy(golden['x']).to(device).requires_grad_(True)
    t = torch.from_numpy(golden['t']).to(device)
    l = crit(x, t)
    if tag == 'bce_none':
        l.backward(torch.from_numpy(golden['map_seed']).to(device))
    else:
        (x.shape[0] * l).backward()
    ref = golden['loss_' + tag]
    np.testing.assert_allclose(l.detach().cpu().numpy(), ref, rtol=2e-5, atol=1e-6)
    dref = golden['dx_' + tag]
    np.testing.assert_allclose(x.grad.cpu().numpy(), dref, rtol=2e-4, atol=3e-6 * np.abs(dref).max())


# ---- uint8 HWC network input: NormalizeImage of lib/augmentations.py:452-460 restated in numpy
def normalize_image_ref(img_u8, scale=1. / 255., mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """x = (x * scale - mean) / std  exactly as the reference writes it (uint8 array * python float -> float64)."""
    x = (img_u8 * float(scale) - np.array(mean, dtype=np.float32)) / np.array(std, dtype=np.float32)
    return x


def check_uint8_input(device, dtype):
    """model(uint8 NHWC batch) == model(float NCHW batch normalised by the reference's formula): eval logits and one
    training step (loss, first-layer weight gradient = the only gradient that sees the input)."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.zf_unet import ZF_UNET
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, size=(2, 128, 160, 3)).astype(np.uint8)
    y = torch.from_numpy((rng.rand(2, 1, 128, 160) > 0.7).astype(np.int64)).to(device)
    xf = torch.from_numpy(np.moveaxis(normalize_image_ref(img), -1, 1).astype(np.float32)).to(device)
    xu = torch.from_numpy(img).to(device)
    res = []
    for x in (xf, xu):
        torch.manual_seed(4)
        m = ZF_UNET(filters=8, dropout_val=0.0).set_compute_dtype(dtype).to(device)
        m.eval()
        with torch.no_grad():
            ev = m(x).clone()
        m.train()
        out = m(x)
        loss = BCEWithLogitsLossAndSmoothJaccard()(out, y)
        (2 * loss).backward()
        res.append((ev.cpu(), out.detach().cpu(), loss.item(), m.conv_224.l1.conv.weight.grad.detach().cpu().clone(),
                    m.conv_7.l2.conv.weight.grad.detach().cpu().clone()))
    (ev_f, out_f, l_f, g1_f, g2_f), (ev_u, out_u, l_u, g1_u, g2_u) = res
    # bf16: both paths round the same normalised value to bf16 (fp32 vs float64 arithmetic before the rounding can
    # move a value across a rounding boundary once in ~1e4 pixels); f32: <= 2 ulp of the input
    tol = 2e-5 if dtype == 'f32' else 2e-3
    s = float(ev_f.abs().max())
    assert float((ev_u - ev_f).abs().max()) <= tol * s, float((ev_u - ev_f).abs().max()) / s
    # training mode: the batch statistics of a 2-image batch amplify the input ulps (ReLU flips; see abi_replay.py)
    assert float((out_u - out_f).abs().max()) <= (1e-3 if dtype == 'f32' else 2e-2) * float(out_f.abs().max())
    assert abs(l_u - l_f) < (1e-5 if dtype == 'f32' else 1e-4)
    # gradients: a tiny random-init net amplifies an input ulp through its small-batch BatchNorms (see abi_replay.py), so
    # the bound is on the relative L2 error per tensor, not per element
    for a, b in ((g1_u, g1_f), (g2_u, g2_f)):
        assert float((a - b).norm() / b.norm()) <= (5e-2 if dtype == 'f32' else 0.2), float((a - b).norm() / b.norm())


def check_inplace_abn_surface(device):
    """InPlaceABN constructor surface (bn.py:50): activation in {leaky_relu, elu, none}, affine False, and the
    inference-mode backward (functions.py:113-116) vs torch BatchNorm2d + activation on the CPU."""
    from lib.modules.abn import InPlaceABN
    import torch.nn.functional as F
    torch.manual_seed(7)
    x = torch.randn(3, 12, 9, 7)
    r = torch.randn(3, 12, 9, 7)
    for activation, affine, train in (('elu', True, True), ('none', False, True), ('leaky_relu', False, True),
                                      ('leaky_relu', True, False)):
        abn = InPlaceABN(12, activation=activation, affine=affine).to(device)
        bn = torch.nn.BatchNorm2d(12, affine=affine)
        with torch.no_grad():
            abn.running_mean.copy_(0.1 * torch.randn(12)); abn.running_var.copy_(0.5 + torch.rand(12))
            bn.running_mean.copy_(abn.running_mean.cpu()); bn.running_var.copy_(abn.running_var.cpu())
            if affine:
                abn.weight.copy_(1 + 0.2 * torch.randn(12)); abn.bias.copy_(0.1 * torch.randn(12))
                bn.weight.copy_(abn.weight.cpu()); bn.bias.copy_(abn.bias.cpu())
        abn.train(train); bn.train(train)
        act = {'elu': F.elu, 'none': lambda t: t, 'leaky_relu': lambda t: F.leaky_relu(t, 0.01)}[activation]
        xa, xb = x.clone().to(device).requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = abn(xa), act(bn(xb))
        (ya * r.to(device)).sum().backward()
        (yb * r).sum().backward()
        tag = (activation, affine, train)
        torch.testing.assert_close(ya.detach().cpu(), yb.detach(), rtol=1e-5, atol=1e-5, msg=str(tag))
        torch.testing.assert_close(xa.grad.cpu(), xb.grad, rtol=1e-4, atol=1e-5, msg=str(tag))
        if affine:
            torch.testing.assert_close(abn.weight.grad.cpu(), bn.weight.grad, rtol=1e-4, atol=1e-5, msg=str(tag))
            torch.testing.assert_close(abn.bias.grad.cpu(), bn.bias.grad, rtol=1e-4, atol=1e-5, msg=str(tag))
        else:
            assert abn.weight is None and abn.bias is None and 'weight' not in abn.state_dict()
        torch.testing.assert_close(abn.running_var.cpu(), bn.running_var, rtol=1e-5, atol=1e-6, msg=str(tag))


def check_inplace_abn_abs_form(device):
    """InPlaceABN(affine_form='abs_eps'): y = leaky(yhat * (|w| + eps) + b), dw = sign(w) * sum(dz * yhat) -- the affine
    form of the inplace_abn API generation that lib/modules/abn/functions.py:81-118 binds -- against a torch restatement,
    with negative and near-zero weights; and the default 'gamma' form against nn.BatchNorm2d on the same weights (the
    two differ there)."""
    import torch.nn.functional as F
    from lib.modules.abn import InPlaceABN
    torch.manual_seed(7)
    C = 10
    x = torch.randn(4, C, 6, 5)
    w0 = torch.tensor([1.0, -0.7, 1e-7, -1e-7, 0.3, -1.5, 2.0, 1.0, -1.0, 0.05])
    b0 = 0.1 * torch.randn(C)
    r = torch.randn(4, C, 6, 5)
    eps = 1e-5

    def restated(xin, w, b, abs_form):
        mean = xin.mean((0, 2, 3), keepdim=True)
        var = xin.var((0, 2, 3), unbiased=False, keepdim=True)
        scale = (w.abs() + eps) if abs_form else w
        return F.leaky_relu((xin - mean) * torch.rsqrt(var + eps) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1), 0.01)

    outs = {}
    for form in ('abs_eps', 'gamma'):
        abn = InPlaceABN(C, affine_form=form).to(device)
        with torch.no_grad():
            abn.weight.copy_(w0)
            abn.bias.copy_(b0)
        xa = x.clone().to(device).requires_grad_(True)
        ya = abn(xa)
        (ya * r.to(device)).sum().backward()
        w = w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        xb = x.clone().requires_grad_(True)
        yb = restated(xb, w, b, form == 'abs_eps')
        (yb * r).sum().backward()
        torch.testing.assert_close(ya.detach().cpu(), yb.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(xa.grad.cpu(), xb.grad, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(abn.weight.grad.cpu(), w.grad, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(abn.bias.grad.cpu(), b.grad, rtol=1e-4, atol=2e-5)
        outs[form] = ya.detach().cpu()
    # the forms are different functions of the same parameters (negative weights flip the sign of the scaled term)
    assert float((outs['abs_eps'] - outs['gamma']).abs().max()) > 0.1
    # at the weight = 1 initialisation they differ by the factor 1 + eps on the normalised term
    a1, a2 = InPlaceABN(C, affine_form='abs_eps').to(device), InPlaceABN(C, affine_form='gamma').to(device)
    d = (a1(x.clone().to(device)) - a2(x.clone().to(device))).abs().max().item()
    assert 0.0 < d < 1e-4
    with pytest.raises(ValueError):
        InPlaceABN(C, affine_form='other')


# ---- bf16 yardstick: the reference's own arithmetic under torch.autocast('cpu', bfloat16) -------------------------------
def grad_cosines(ga, gb, names):
    """(global cosine, (worst tensor name, its cosine)) over the named gradient tensors."""
    worst = ('', 1.0)
    num = da = db = 0.0
    for n in names:
        a, b = ga[n].double().reshape(-1), gb[n].double().reshape(-1)
        ab, aa, bb = float((a * b).sum()), float((a * a).sum()), float((b * b).sum())
        c = ab / max(np.sqrt(aa * bb), 1e-300)
        if c < worst[1]:
            worst = (n, c)
        num, da, db = num + ab, da + aa, db + bb
    return num / np.sqrt(da * db), worst


def autocast_yardstick(sd_fn, x, y, loss_name, drop):
    """The oracle step in fp32 and under bf16 autocast on the same weights / batch / dropout draw.
    -> dict(dloss, diou, cos, worst, flipped) of autocast relative to fp32, plus the fp32 results for further use."""
    from oracle import train_step_ref
    l32, o32, g32 = train_step_ref.loss_and_grads(sd_fn(), x, y, loss_name, drop=drop)
    l16, o16, g16 = train_step_ref.loss_and_grads(sd_fn(), x, y, loss_name, drop=drop, autocast=True)
    names = [n for n in g32 if n.endswith('conv.weight') or n == 'conv_final.weight']
    cos, worst = grad_cosines(g16, g32, names)
    return dict(dloss=float(l16 - l32), diou=float(losses_ref.jaccard_score(o16, y) - losses_ref.jaccard_score(o32, y)),
                cos=cos, worst=worst, flipped=float(((o16 > 0) != (o32 > 0)).float().mean()),
                fp32=(float(l32), o32, g32), names=names)


def blob_batch(B, S, seed):
    """A learnable synthetic segmentation task: 2-5 filled rectangles / discs per image on a noisy background; the mask
    is their union, the image shows them as brighter, differently tinted regions."""
    rng = np.random.RandomState(seed)
    x = 0.35 * rng.randn(B, 3, S, S).astype(np.float32)
    y = np.zeros((B, 1, S, S), dtype=np.int64)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        for _ in range(rng.randint(2, 6)):
            cy, cx, r = rng.randint(0, S), rng.randint(0, S), rng.randint(S // 12, S // 4)
            m = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r) if rng.rand() < 0.5 else \
                ((abs(yy - cy) <= r) & (abs(xx - cx) <= r // 2 + 1))
            y[b, 0][m] = 1
            x[b][:, m] += (0.8 + 0.4 * rng.rand(3, 1)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y)


def check_linknet_abs_eps_form(device, dtype='f32', size=64):
    """LinkNet34 whose InPlaceABN layers use the backend's affine form (lib/modules/abn/functions.py:94,112,118: scale
    |w| + eps, weight gradient signed) with NEGATIVE and near-zero weights, against the same network in the gamma form --
    the path pinned to the reference golden -- carrying |w| + eps as its weights: equal logits, equal gradients but for the
    sign on the InPlaceABN weights."""
    import warnings
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.linknet import LinkNet34
    from lib.modules.abn import InPlaceABN
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(21)
        a = LinkNet34(num_classes=1)
        torch.manual_seed(21)
        b = LinkNet34(num_classes=1)
    a.finaldrop1.p = b.finaldrop1.p = 0.0
    gen = torch.Generator().manual_seed(5)
    signs = {}
    for (n, ma), (_, mb) in zip(a.named_modules(), b.named_modules()):
        if isinstance(ma, InPlaceABN):
            w = (torch.rand(ma.num_features, generator=gen) + 0.3) * torch.where(torch.rand(ma.num_features, generator=gen) > 0.5, 1.0, -1.0)
            w[0] = 1e-7                                    # near zero: the gamma form would lose the channel
            ma.affine_form = 'abs_eps'
            with torch.no_grad():
                ma.weight.copy_(w)
                mb.weight.copy_(w.abs() + ma.eps)
            signs[n + '.weight'] = torch.where(w > 0, 1.0, -1.0)
    x = torch.randn(2, 3, size, size + 32, generator=gen)
    y = (torch.rand(2, 1, size, size + 32, generator=gen) > 0.7).long()
    outs, grads = [], []
    for m in (a, b):
        m.set_compute_dtype(dtype)
        m.to(device).train()
        out = m(x.to(device))
        loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.to(device))
        (2 * loss).backward()
        outs.append(out.detach().cpu())
        grads.append({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    tol = 1e-5 if dtype == 'f32' else 2e-2
    assert float((outs[0] - outs[1]).abs().max()) <= tol * float(outs[1].abs().max())
    gmax = max(float(g.abs().max()) for g in grads[1].values())
    for n, gb in grads[1].items():
        ga = grads[0][n]
        if n in signs:
            gb = gb * signs[n]
        assert float((ga - gb).abs().max()) <= (1e-4 if dtype == 'f32' else 5e-2) * max(float(gb.abs().max()), 1e-3 * gmax), n


def check_executor_fused_reduce(device, size=32):
    """segnb.net: conv -> BatchNorm -> ReLU -> conv.  The second convolution is the ONLY consumer of the first one's activated
    output, so its data gradient does that layer's BatchNorm-backward reduction in its epilogue (segnb_conv_fprop_bnreduce via
    net._data_gradient, the wiring of linknet.py:41-62's basic blocks) -- against the same step with every reduction as a pass of
    its own (Tape.fuse_reduce off).  Same kernels' arithmetic, bf16: gradients agree to rounding of the sums."""
    from torch import nn
    from segnb import net as _net
    from segnb import _native as nv
    from segnb import convplan as cp
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard

    class Toy(_net.HipNet):
        def __init__(self):
            super(Toy, self).__init__()
            self.c1, self.b1 = nn.Conv2d(3, 32, 3, padding=1), nn.BatchNorm2d(32)
            self.c2, self.b2 = nn.Conv2d(32, 32, 3, padding=1), nn.BatchNorm2d(32)
            self.final = nn.Conv2d(32, 1, 1)
            self.fused_seen = 0
            self._init_engine(3)

        def _build(self, tape, x, dlogits):
            a1 = _net.conv_unit(tape, x, self.c1.weight, self.c1.bias, [(3, cp.pad8(3))], bn=self.b1, act=nv.ACT_RELU, tag='c1')
            a2 = _net.conv_unit(tape, a1, self.c2.weight, self.c2.bias, [(32, 32)], bn=self.b2, act=nv.ACT_RELU, tag='c2')
            self._a1 = a1
            return _net.head_1x1(tape, a2, self.final.weight, self.final.bias, dlogits)

    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, size, size, generator=gen)
    y = (torch.rand(2, 1, size, size, generator=gen) > 0.6).long()
    res = []
    keep = _net.Tape.fuse_reduce
    try:
        for fuse in (True, False):
            _net.Tape.fuse_reduce = fuse
            torch.manual_seed(11)
            m = Toy().to(device).train()
            taken = []
            orig = _net.ConvOp.dgrad

            def spy(self, dyv, dxv, bn_reduce=None, _orig=orig, _taken=taken):
                _taken.append(bn_reduce is not None)
                return _orig(self, dyv, dxv, bn_reduce=bn_reduce)
            _net.ConvOp.dgrad = spy
            try:
                out = m(x.to(device))
                loss = BCEWithLogitsLossAndSmoothJaccard()(out, y.to(device))
                (2 * loss).backward()
            finally:
                _net.ConvOp.dgrad = orig
            assert any(taken) == fuse, (fuse, taken)            # the fused launch was (not) taken
            res.append((out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}))
    finally:
        _net.Tape.fuse_reduce = keep
    (oa, ga), (ob, gb) = res
    assert torch.equal(oa, ob)
    gmax = max(float(g.abs().max()) for g in gb.values())
    for n in gb:
        assert float((ga[n] - gb[n]).abs().max()) <= 2e-2 * max(float(gb[n].abs().max()), 1e-3 * gmax), n
