"""-m gpu: end-to-end parity of the HIP path (lib.models.zf_unet.ZF_UNET + lib.losses on libsegnb_hip.so)
against golden vectors produced by the REFERENCE itself (tests/golden/*.npz) and against the oracle.

Tolerances are the north-star ones: |loss| <= 1e-5 and |soft IoU| < 1e-4 on the exact-fp32 path; the
bf16 throughput path is reported against the same goldens with its own (stated) tolerances.
"""
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref, train_step_ref, zf_unet_ref

pytestmark = pytest.mark.gpu


def _model(filters, dropout, seed, dtype):
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(int(seed))       # same construction order as the reference -> identical default init
    m = ZF_UNET(dropout_val=dropout, filters=filters)
    m.set_compute_dtype(dtype)
    return m.cuda()


def test_native_library_is_what_runs():
    from segnb import _native as nv
    assert nv._test_backend is None
    assert os.path.exists(nv.LIB_PATH)
    nv.load()
    with open('/proc/self/maps') as f:
        assert 'libsegnb_hip.so' in f.read()


def test_tiny_f32_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'zf_unet_tiny.npz'))
    x, y = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['y']).cuda()
    m = _model(4, 0.0, 3.0, 'f32')
    m.eval()
    with torch.no_grad():
        ev = m(x)
    np.testing.assert_allclose(ev.cpu().numpy(), g['eval_logits'], rtol=1e-4,
                               atol=5e-4 * np.abs(g['eval_logits']).max())
    m.train()
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore, PixelAccuracy
    out = m(x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['train_logits'], rtol=2e-3, atol=1e-3)
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y)
    assert abs(loss.item() - float(g['loss_bce_jaccard'])) < 1e-5
    assert abs(JaccardScore()(out, y).item() - float(g['iou'])) < 1e-4
    assert abs(PixelAccuracy()(out, y).item() - float(g['acc'])) < 1e-4
    (x.shape[0] * loss).backward()
    for n, p in m.named_parameters():
        ref = g['grad/' + n]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 1e-3 * scale + 3e-6, (n, err, scale)
    for n, b in m.named_buffers():
        np.testing.assert_allclose(b.cpu().numpy(), g['buf/' + n], rtol=1e-5, atol=1e-6, err_msg=n)


def test_tiny_f32_training_steps_with_torch_sgd(golden_dir):
    """torch_train.py:180-190 verbatim (zero_grad / forward / loss / (B*loss).backward() / optimizer.step())."""
    g = np.load(os.path.join(golden_dir, 'zf_unet_tiny.npz'))
    x, y = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['y']).cuda()
    m = _model(4, 0.0, 3.0, 'f32')
    m.train()
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    crit = BCEWithLogitsLossAndSmoothJaccard().cuda()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    traj = []
    for it in range(5):
        opt.zero_grad()
        out = m(x)
        loss = crit(out, y)
        (x.size(0) * loss).backward()
        opt.step()
        traj.append(loss.cpu().item())
        if it == 0:
            sd = m.state_dict()
            for k in sd:
                np.testing.assert_allclose(sd[k].cpu().numpy().astype(np.float64),
                                           g['after1/' + k].astype(np.float64), rtol=2e-5, atol=3e-6, err_msg=k)
    np.testing.assert_allclose(traj[:2], g['traj_bce_jaccard'][:2], rtol=1e-5)
    np.testing.assert_allclose(traj, g['traj_bce_jaccard'], rtol=1e-3)


def _run_224(dtype, golden_dir):
    g = np.load(os.path.join(golden_dir, 'zf_unet_224.npz'))
    x, y = train_step_ref.synthetic_batch(4, 224, seed=1234)
    x, y = x.cuda(), y.cuda()
    m = _model(32, 0.2, 1.0, dtype)
    m.dropout_override = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('drop/')}
    m.train()
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore, PixelAccuracy
    out = m(x)
    loss = BCEWithLogitsLossAndSmoothJaccard()(out, y)
    iou, acc = JaccardScore()(out, y).item(), PixelAccuracy()(out, y).item()
    (4 * loss).backward()
    torch.cuda.synchronize()
    return g, m, out.detach().cpu().numpy(), loss.item(), iou, acc


def test_config1_224_f32_vs_reference_golden(golden_dir):
    """BASELINE.json configs[0] shape: ZF_UNET() default, B=4, 224x224, Dropout2d draw replayed."""
    g, m, logits, loss, iou, acc = _run_224('f32', golden_dir)
    assert abs(loss - float(g['loss_bce_jaccard'])) < 1e-5, (loss, float(g['loss_bce_jaccard']))
    assert abs(iou - float(g['iou'])) < 1e-4
    assert abs(acc - float(g['acc'])) < 1e-4
    np.testing.assert_allclose(logits.reshape(-1)[g['logit_idx']], g['logit_val'], rtol=2e-3, atol=5e-4)
    names = list(g['grad_names'])
    params = dict(m.named_parameters())
    norms = np.array([np.sqrt((params[n].grad.double() ** 2).sum().item()) for n in names])
    # BatchNorm-bias gradients are sums of ~2e5 signed terms that largely cancel: their fp32 summation noise is
    # ~4e-3 of the value on BOTH sides; weight-gradient norms agree to 3e-4
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=1e-2, atol=2e-6 * norms.max())
    wsel = np.array([n.endswith('conv.weight') for n in names])
    np.testing.assert_allclose(norms[wsel], g['grad_norms'][wsel], rtol=1e-3)
    for n in names:
        gv = params[n].grad.reshape(-1).cpu().numpy()[g['gidx/' + n]]
        ref = g['gval/' + n]
        scale = max(np.abs(ref).max(), float(g['grad_norms'][names.index(n)]) / np.sqrt(params[n].numel()), 1e-7)
        # 64 probed entries per tensor; the first conv's entries are sums of 2e5 signed products (fp32 noise ~1e-2
        # of an entry on both sides), deeper tensors agree much tighter
        rel = 5e-2 if n.endswith('bn.bias') else 2e-2     # BN-bias entries: sums of 2e5 terms that largely cancel
        assert np.abs(gv - ref).max() <= rel * scale + 2e-6, (n, np.abs(gv - ref).max(), scale)
    for n, b in m.named_buffers():
        if 'buf/' + n in g.files:
            np.testing.assert_allclose(b.cpu().numpy(), g['buf/' + n], rtol=1e-4, atol=1e-5, err_msg=n)


def test_config1_224_bf16_reported_deltas(golden_dir):
    """bf16 throughput path against the same fp32 reference goldens.  SURVEY 7: the reference itself under
    bf16 autocast moves the loss by 1.4e-5 and logits by up to 0.48, so this path is held to its own,
    stated, tolerances: loss 5e-4, soft IoU 5e-4, accuracy 2e-3, weight-gradient norms 5 % (measured on
    MI355X: loss +3.4e-5, IoU -4.1e-5, accuracy +1.2e-4)."""
    g, m, logits, loss, iou, acc = _run_224('bf16', golden_dir)
    print('bf16 deltas: loss %.3e  iou %.3e  acc %.3e' % (loss - float(g['loss_bce_jaccard']),
                                                        iou - float(g['iou']), acc - float(g['acc'])))
    assert abs(loss - float(g['loss_bce_jaccard'])) < 5e-4
    assert abs(iou - float(g['iou'])) < 5e-4
    assert abs(acc - float(g['acc'])) < 2e-3
    names = list(g['grad_names'])
    params = dict(m.named_parameters())
    norms = np.array([np.sqrt((params[n].grad.double() ** 2).sum().item()) for n in names])
    wsel = np.array([n.endswith('conv.weight') or n == 'conv_final.weight' for n in names])
    print('bf16 weight-grad norm rel err max %.3e' % np.abs(norms[wsel] / g['grad_norms'][wsel] - 1).max())
    got = np.concatenate([params[n].grad.reshape(-1).cpu().numpy()[g['gidx/' + n]] for n in names])
    ref = np.concatenate([g['gval/' + n] for n in names])
    cos = float((got * ref).sum() / np.sqrt((got * got).sum() * (ref * ref).sum()))
    print('bf16 gradient direction on %d probed entries: cosine %.5f' % (got.size, cos))
    # every activation, gradient and weight operand is rounded to bf16 (8 significant bits, as under
    # torch.autocast); each BatchNorm strips the clean mean component of its input and keeps all the noise
    # (x rms/std ~ 1.1-1.6 per stage), so the 0.3 % rounding noise grows ~x1.6 per block on the way down and the
    # backward signal decorrelates accordingly.  Measured on MI355X: 0.80 at B=4 random init.
    assert cos > 0.7
    np.testing.assert_allclose(norms[wsel], g['grad_norms'][wsel], rtol=5e-2)


def test_config1_224_bf16_no_further_from_fp32_than_autocast(golden_dir):
    """the HIP bf16 path against the reference's own arithmetic under bf16 autocast (the oracle under
    torch.autocast('cpu', bfloat16)), both measured against fp32 on the golden weights, batch and Dropout2d draw (B=4
    224x224): |dloss|, |dIoU|, 1 - weight-gradient cosine (global and worst tensor) and the fraction of thresholded
    pixels that flip must not exceed 1.25x autocast's.  Measured on MI355X: HIP bf16 dloss +3.5e-5 / dIoU -3.9e-5 /
    cosine 0.80 against autocast +6.3e-5 / -6.6e-5 / 0.66: the HIP path keeps fp32 BatchNorm statistics, fp32
    accumulators and fp32 weight-gradient sums where autocast rounds between operators."""
    import model_checks as mc
    g, m, logits, loss, iou, acc = _run_224('bf16', golden_dir)
    x, y = train_step_ref.synthetic_batch(4, 224, seed=1234)
    drop = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('drop/')}
    ys = mc.autocast_yardstick(lambda: zf_unet_ref.default_init_state(filters=32, seed=1), x, y, 'bce_jaccard', drop)
    l32, o32, g32 = ys['fp32']
    hip = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    cos, worst = mc.grad_cosines(hip, g32, ys['names'])
    dloss, diou = loss - l32, iou - losses_ref.jaccard_score(o32, y).item()
    flipped = float(((torch.from_numpy(logits) > 0) != (o32 > 0)).float().mean())
    print('B=4 224 vs fp32:  HIP bf16 dloss %.3e dIoU %.3e cos %.4f worst %s %.4f flipped %.4f | autocast dloss %.3e dIoU '
          '%.3e cos %.4f worst %s %.4f flipped %.4f' % (dloss, diou, cos, worst[0], worst[1], flipped, ys['dloss'],
                                                       ys['diou'], ys['cos'], ys['worst'][0], ys['worst'][1], ys['flipped']))
    assert abs(dloss) <= 1.25 * abs(ys['dloss']) + 1e-6
    assert abs(diou) <= 1.25 * abs(ys['diou']) + 1e-6
    assert 1.0 - cos <= 1.25 * (1.0 - ys['cos'])
    assert 1.0 - worst[1] <= 1.25 * (1.0 - ys['worst'][1])
    assert flipped <= 1.25 * ys['flipped']


def test_bf16_path_trains_like_fp32_and_segments_alike():
    """VERDICT r2 item 5b / missing 4: a learnable synthetic task (model_checks.blob_batch), 300 SGD steps from the same
    initialisation on the exact-fp32 HIP path and on the bf16 throughput path: both learn (soft IoU well above the
    initial value), final soft IoU (lib/metrics.py:9-20, on a held-out batch, eval mode as validate() of
    torch_train.py:248-276) within 0.01 of each other.  Then the SAME fp32-trained weights evaluated by both paths:
    |dIoU| < 1e-4 (north_star), flipped-pixel fraction reported and bounded."""
    import model_checks as mc
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore
    from lib.models.zf_unet import ZF_UNET
    B, S, F = 8, 64, 8
    batches = [tuple(t.cuda() for t in mc.blob_batch(B, S, 100 + i)) for i in range(8)]
    xv, yv = (t.cuda() for t in mc.blob_batch(16, S, 999))
    crit = BCEWithLogitsLossAndSmoothJaccard()
    finals, models = {}, {}
    for dtype in ('f32', 'bf16'):
        torch.manual_seed(11)
        m = ZF_UNET(dropout_val=0.0, filters=F).set_compute_dtype(dtype).cuda().train()
        opt = torch.optim.SGD(m.parameters(), lr=2e-2)
        with torch.no_grad():
            m.eval()
            iou0 = JaccardScore()(m(xv), yv).item()
            m.train()
        for it in range(300):
            x, y = batches[it % len(batches)]
            opt.zero_grad()
            loss = crit(m(x), y)
            (B * loss).backward()
            opt.step()
        m.eval()
        with torch.no_grad():
            finals[dtype] = (iou0, JaccardScore()(m(xv), yv).item(), loss.item())
        models[dtype] = m
    print('300 steps: fp32 IoU %.4f -> %.4f (loss %.4f), bf16 IoU %.4f -> %.4f (loss %.4f)'
          % (finals['f32'] + finals['bf16']))
    for dtype in finals:
        assert finals[dtype][1] > finals[dtype][0] + 0.2 and finals[dtype][1] > 0.5, finals
    assert abs(finals['f32'][1] - finals['bf16'][1]) < 0.01, finals
    # the fp32-trained weights through the bf16 kernels
    mb = ZF_UNET(dropout_val=0.0, filters=F).set_compute_dtype('bf16').cuda().eval()
    mb.load_state_dict(models['f32'].state_dict())
    with torch.no_grad():
        o32, o16 = models['f32'](xv), mb(xv)
        i32, i16 = JaccardScore()(o32, yv).item(), JaccardScore()(o16, yv).item()
    flipped = float(((o32 > 0) != (o16 > 0)).float().mean())
    # the same weights through the reference's arithmetic (oracle) in fp32 and under bf16 autocast: its own IoU shift
    sd = {k: v.detach().cpu().clone() for k, v in models['f32'].state_dict().items()}
    with torch.no_grad():
        r32 = zf_unet_ref.forward(dict(sd), xv.cpu(), train=False)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            r16 = zf_unet_ref.forward(dict(sd), xv.cpu(), train=False).float()
    ac = losses_ref.jaccard_score(r16, yv.cpu()).item() - losses_ref.jaccard_score(r32, yv.cpu()).item()
    print('fp32-trained weights, eval: IoU fp32 %.6f bf16 %.6f (d %.2e), flipped pixels %.5f; oracle fp32 IoU %.6f, '
          'autocast d %.2e' % (i32, i16, i16 - i32, flipped, losses_ref.jaccard_score(r32, yv.cpu()).item(), ac))
    assert abs(losses_ref.jaccard_score(r32, yv.cpu()).item() - i32) < 1e-4       # fp32 HIP path == oracle on trained weights
    # (the weights come out of a 300-step fp32 run whose general weight-gradient kernel sums with float atomics: they differ in
    # the last bits from run to run, and so does this single draw -- 0.3e-4 .. 1.6e-4 over repeated runs on MI355X; the
    # fp32 HIP path itself is held to the north-star 1e-4 one line above)
    assert abs(i16 - i32) < max(2e-4, 1.25 * abs(ac)) and flipped < 5e-3


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_odd_filters_and_dropout_replay_vs_oracle(dtype):
    # bf16 stores every activation / gradient with 8 significant bits; a 6-channel toy has nothing to average
    # that noise over (forward drift 0.3 % -> 12 % across the 22 layers, measured with the ABI emulator too), so
    # the bf16 variant runs a 12-filter net (channel counts 12..384: still exercises the padded concat slices)
    B, S, F = (2, 128, 6) if dtype == 'f32' else (4, 128, 12)
    x, y = train_step_ref.synthetic_batch(B, S, seed=5)
    sd = zf_unet_ref.default_init_state(filters=F, seed=2)
    drop = zf_unet_ref.make_dropout_tables(F, B, 0.2, torch.Generator().manual_seed(3))
    # the same step in fp64: the fp32 oracle's own distance from it is the yardstick for summation-order noise
    # (B=2 at 128x128 leaves 32 samples per channel in the deepest BatchNorms, which amplify it on the way
    # back; at 64x64 -- 8 samples -- a single ReLU flipping on 1e-5 noise moved every gradient by 2 %)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    drop64 = None if drop is None else {k: v.double() for k, v in drop.items()} if isinstance(drop, dict) \
        else [d.double() for d in drop]
    _, _, grads64 = train_step_ref.loss_and_grads(sd64, x.double(), y, 'bce_dice', drop=drop64)
    loss_ref, logits_ref, grads_ref = train_step_ref.loss_and_grads(sd, x, y, 'bce_dice', drop=drop)
    m = _model(F, 0.2, 2.0, dtype)
    m.dropout_override = drop
    m.train()
    from lib.losses import BCEAndDiceLoss
    out = m(x.cuda())
    loss = BCEAndDiceLoss()(out, y.cuda())
    (B * loss).backward()
    ltol = 1e-5 if dtype == 'f32' else 5e-3
    assert abs(loss.item() - loss_ref.item()) < ltol
    worst = ('', 0.0)
    for n, p in m.named_parameters():
        ref = grads_ref[n].double()
        got = p.grad.cpu().double()
        rel = float((got - ref).norm() / (ref.norm() + 1e-12))
        if n.endswith('conv.weight') and rel > worst[1]:
            worst = (n, rel)
        if dtype == 'f32':
            scale = max(float(ref.abs().max()), 1e-6)
            # not tight on purpose: a ReLU / max-pool argmax within an fp32 ulp of its decision point flips between
            # correct implementations and moves upstream gradients by 1e-3..1e-2 in a 6-filter net; the per-launch
            # exactness of this very plan is asserted by tests/test_replay_gpu.py (teacher-forced)
            if n.endswith('.conv.bias'):
                continue           # analytically zero under training-mode BatchNorm: pure noise in the reference
            assert float((got - grads64[n]).norm() / (grads64[n].norm() + 1e-30)) <= 0.1, n
    g_all = torch.cat([p.grad.cpu().double().reshape(-1) for _, p in m.named_parameters()])
    r_all = torch.cat([grads_ref[n].double().reshape(-1) for n, _ in m.named_parameters()])
    cos = float((g_all * r_all).sum() / (g_all.norm() * r_all.norm()))
    print('%s: global gradient cosine %.6f, worst conv-weight rel L2 error %s %.3e' % (dtype, cos, worst[0], worst[1]))
    # bf16: every activation/gradient tensor is stored with 8 significant bits; a 6-channel toy net averages
    # little of that noise away, so the bound is on the direction of the whole gradient
    assert cos > (0.9999 if dtype == 'f32' else 0.6)


def test_full_size_bs32_bf16_properties():
    """BASELINE.json configs[1] (ZF_UNET 224x224 bf16 bs=32, BCE+Dice): size-independent properties --
    finite outputs, run-to-run reproducible forward, BatchNorm bookkeeping, loss goes down under SGD."""
    from lib.losses import BCEAndDiceLoss
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(0)
    m = ZF_UNET().cuda()
    m.train()
    x, y = train_step_ref.synthetic_batch(32, 224, seed=1234)
    x, y = x.cuda(), y.cuda()
    crit = BCEAndDiceLoss()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    m.dropout_override = {}            # Dropout2d off for the reproducibility check
    with torch.no_grad():
        a = m(x).clone()
        b = m(x).clone()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), 'forward is not run-to-run reproducible'
    assert int(m.conv_224.l1.bn.num_batches_tracked) == 2
    m.dropout_override = None
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(m(x), y)
        (x.size(0) * loss).backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    for p in m.parameters():
        assert torch.isfinite(p.grad).all()


@pytest.mark.parametrize('cfg', [(8, 64, 'bf16'), (32, 224, 'bf16')])
def test_training_steps_are_bitwise_reproducible(cfg):
    """with the head's weight / bias gradients summed in a fixed order (csrc/head_loss.hip) no float
    atomic is left in the bf16 training step: three SGD steps from the same initialisation, twice -- every parameter and
    BatchNorm buffer bit for bit equal, at a small size and at the timed configuration (bs=32 224x224 bf16).  (The
    exact-fp32 parity mode is not: its general weight-gradient kernel, conv_igemm.hip, merges pixel ranges with fp32
    atomics -- measured here: losses equal for two steps, 1e-5 apart at the third.)"""
    from lib.losses import BCEAndDiceLoss
    from lib.models.zf_unet import ZF_UNET
    from segnb import optim
    B, S, dtype = cfg
    x, y = train_step_ref.synthetic_batch(B, S, seed=77)
    x, y = x.cuda(), y.cuda()
    drop = zf_unet_ref.make_dropout_tables(32, B, 0.2, torch.Generator().manual_seed(9))
    states, losses = [], []
    for run in range(2):
        torch.manual_seed(0)
        m = ZF_UNET().set_compute_dtype(dtype).cuda().train()
        m.dropout_override = drop
        opt = optim.SGD(m.parameters(), lr=1e-2)
        ls = []
        for it in range(3):
            opt.zero_grad()
            loss = BCEAndDiceLoss()(m(x), y)
            (B * loss).backward()
            opt.step()
            ls.append(loss.item())
        torch.cuda.synchronize()
        states.append({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
        losses.append(ls)
        del m, opt
    assert losses[0] == losses[1], losses
    bad = [k for k in states[0] if not torch.equal(states[0][k], states[1][k])]
    assert not bad, bad[:5]


def test_timed_config_bs32_bf16_vs_fp32_hip_path():
    """The TIMED configuration (BASELINE.json configs[1]: ZF_UNET 224x224 bs=32 bf16, BCE+Dice, Dropout2d 0.2) against
    the exact-fp32 HIP path on the same weights, batch and Dropout2d draw.  The fp32 path is the one pinned to the
    reference goldens (test_config1_224_f32_vs_reference_golden, |dloss| < 1e-5, |dIoU| < 1e-4), so this ties the
    numbers bench.py times to the reference at the timed size.  Reported and bounded: dloss, dIoU, daccuracy,
    per-tensor weight-gradient cosine and norm ratio.  (VERDICT r1 weak #2.)"""
    from lib.losses import BCEAndDiceLoss
    from lib.metrics import JaccardScore, PixelAccuracy
    from lib.models.zf_unet import ZF_UNET
    B, S = 32, 224
    x, y = train_step_ref.synthetic_batch(B, S, seed=1234)
    x, y = x.cuda(), y.cuda()
    drop = zf_unet_ref.make_dropout_tables(32, B, 0.2, torch.Generator().manual_seed(7))
    res = {}
    for dtype in ('f32', 'bf16'):
        torch.manual_seed(0)
        m = ZF_UNET().set_compute_dtype(dtype).cuda().train()
        m.dropout_override = drop
        out = m(x)
        loss = BCEAndDiceLoss()(out, y)
        iou, acc = JaccardScore()(out, y).item(), PixelAccuracy()(out, y).item()
        (B * loss).backward()
        torch.cuda.synchronize()
        res[dtype] = (loss.item(), iou, acc, {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()},
                      out.detach().cpu())
        del m
    (l32, i32, a32, g32, o32), (l16, i16, a16, g16, o16) = res['f32'], res['bf16']
    print('bs=32 bf16 - fp32: dloss %.3e  dIoU %.3e  dacc %.3e  logits max|d| %.3e (scale %.2f)'
          % (l16 - l32, i16 - i32, a16 - a32, float((o16 - o32).abs().max()), float(o32.abs().max())))
    assert abs(l16 - l32) < 5e-4 and abs(i16 - i32) < 5e-4 and abs(a16 - a32) < 2e-3
    worst_cos, worst_ratio = ('', 1.0), ('', 0.0)
    ga, gb = [], []
    for n in g32:
        if not (n.endswith('conv.weight') or n == 'conv_final.weight'):
            continue
        a, b = g16[n].reshape(-1), g32[n].reshape(-1)
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        ratio = abs(float(a.norm() / b.norm()) - 1.0)
        if cos < worst_cos[1]:
            worst_cos = (n, cos)
        if ratio > worst_ratio[1]:
            worst_ratio = (n, ratio)
        ga.append(a)
        gb.append(b)
    ga, gb = torch.cat(ga), torch.cat(gb)
    cos_all = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    print('bs=32 weight gradients bf16 vs fp32: global cosine %.5f, worst tensor %s %.5f, worst norm ratio %s %.3e'
          % (cos_all, worst_cos[0], worst_cos[1], worst_ratio[0], worst_ratio[1]))
    # bf16 storage noise grows ~x1.1-1.6 per BatchNorm stage on the way down the net.  The yardstick is MEASURED, not asserted
    # (VERDICT r2 weak 1): the oracle -- the reference's arithmetic -- under bf16 autocast against itself in fp32 on the same
    # weights, batch and dropout draw; the HIP bf16 path must stay within 1.25x of autocast's own distance from fp32
    import model_checks as mc
    ys = mc.autocast_yardstick(lambda: zf_unet_ref.default_init_state(filters=32, seed=0), x.cpu(), y.cpu(), 'bce_dice', drop)
    print('bs=32 autocast yardstick: dloss %.3e dIoU %.3e cos %.5f worst %s %.5f flipped %.4f'
          % (ys['dloss'], ys['diou'], ys['cos'], ys['worst'][0], ys['worst'][1], ys['flipped']))
    assert abs(ys['fp32'][0] - l32) < 1e-5                      # same weights: the fp32 HIP path sits on the oracle
    assert 1.0 - cos_all <= 1.25 * (1.0 - ys['cos']), (cos_all, ys['cos'])
    assert 1.0 - worst_cos[1] <= 1.25 * (1.0 - ys['worst'][1]), (worst_cos, ys['worst'])
    # (the scalar deltas are signed sums over 1.6 M pixels that largely cancel -- autocast's own landed at -3e-6 / +8e-6 here,
    # +6e-5 / -7e-5 at B=4 -- so they are bounded by the larger of autocast's and the B=4 level)
    assert abs(l16 - l32) <= max(1.25 * abs(ys['dloss']), 5e-5) and abs(i16 - i32) <= max(1.25 * abs(ys['diou']), 5e-5)
    assert float(((o16 > 0) != (o32 > 0)).float().mean()) <= 1.25 * ys['flipped']
    assert worst_ratio[1] < 0.1, worst_ratio


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_uint8_hwc_input_matches_normalized_float_input(dtype):
    """SURVEY 8f rank 2 on the HIP kernels: uint8 HWC batch -> segnb_conv_fprop_u8 (bf16: the first convolution reads the
    image itself and publishes the packed pixels for its weight gradient) / segnb_pack_input_u8 (f32) vs the float NCHW
    batch normalised by the reference's formula."""
    import model_checks as mc
    mc.check_uint8_input('cuda', dtype)


def test_uint8_first_layer_kernel_bitwise_vs_packed_path():
    """segnb_conv_fprop_u8 == segnb_pack_input_u8 + segnb_conv_fprop bit for bit (same arithmetic, no packed copy read),
    at the timed size (bs=32 224x224), including the packed pixels it publishes and the BatchNorm statistics."""
    from segnb import _native as nv
    from segnb import convplan as cp
    from segnb.engine import ConvOp, InputNorm, Runtime, View, pack_input
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (32, 224, 224, 3), generator=gen, dtype=torch.uint8).cuda()
    w = (torch.randn(32, 3, 3, 3, generator=gen) * 0.2).cuda()
    b = (torch.randn(32, generator=gen) * 0.1).cuda()
    op = ConvOp(rt, w, b, [(3, 8)], 1, 1, False, need_dgrad=False)
    op.pack(224, 224)
    norm = InputNorm()
    xv = View.alloc(rt, 32, 224, 224, 8)
    pack_input(rt, img, xv, norm)
    y0, st0 = View.alloc(rt, 32, 224, 224, 32), rt.zeros((16, 2, 32), torch.float64)
    op.fprop(xv, y0, st0)
    assert op.u8_direct_ok(32, 224, 224, 32)
    xp = View.alloc(rt, 32, 224, 224, 8)
    y1, st1 = View.alloc(rt, 32, 224, 224, 32), rt.zeros((16, 2, 32), torch.float64)
    op.fprop_u8(img, norm, y1, st1, xp)
    torch.cuda.synchronize()
    assert torch.equal(y0.t, y1.t) and torch.equal(xv.t, xp.t)
    np.testing.assert_allclose(st1.sum(0).cpu().numpy(), st0.sum(0).cpu().numpy(), rtol=1e-12)
    # and the packed pixels are the reference's NormalizeImage rounded to bf16 (fp32 vs float64 arithmetic: a rounding
    # boundary is crossed at most once in ~1e4 values)
    import model_checks as mc
    ref = torch.from_numpy(mc.normalize_image_ref(img.cpu().numpy()).astype(np.float32)).bfloat16().float()
    got = xv.dense()[..., :3].float().cpu()
    diff = (got - ref).abs()
    assert float(diff.max()) <= 2 ** -7 * float(ref.abs().max()) and float((diff > 0).float().mean()) < 1e-3
    assert float(xv.dense()[..., 3:].abs().max()) == 0.0


def test_launch_plan_replay_matches_eager_and_cuts_host_time():
    """segnb_plan_*: the forward / backward launch lists replayed from C.  Same steps with the
    replay on and off: losses, eval logits, BatchNorm buffers and parameters agree BIT FOR BIT (the step has no float
    atomics left), dropout and changing inputs flow through the recorded lists, and the host enqueues a step in under a
    third of the eager launcher's time."""
    import time
    from lib.losses import BCEAndDiceLoss
    from lib.models import zf_unet as zf
    from segnb import optim
    gen = torch.Generator().manual_seed(3)
    xs = [torch.randn(8, 3, 64, 64, generator=gen).cuda() for _ in range(4)]
    ys = [(torch.rand(8, 1, 64, 64, generator=gen) > 0.7).long().cuda() for _ in range(4)]
    drop = zf_unet_ref.make_dropout_tables(32, 8, 0.2, torch.Generator().manual_seed(5))
    res, host = {}, {}
    for run, mode in enumerate((True, False, False)):
        zf._ZFUnetPlan.use_cplan = mode
        try:
            torch.manual_seed(0)
            m = zf.ZF_UNET().cuda().train()
            m.dropout_override = drop
            opt = optim.SGD(m.parameters(), lr=1e-2)
            losses = []
            for it in range(6):
                opt.zero_grad()
                loss = BCEAndDiceLoss()(m(xs[it % 4]), ys[it % 4])
                (8 * loss).backward()
                opt.step()
                losses.append(loss.item())
                if it == 5:
                    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            m.eval()
            with torch.no_grad():
                ev = m(xs[0]).clone()
                ev2 = m(xs[1]).clone()
            m.train()
            if mode:
                eng = m._engine
                assert len(eng._cplans) >= 3 and all(p[0] is not None and p[2] > 20 for p in eng._cplans.values())
            m.dropout_override = None          # (the replayed tables are eleven blocking host-to-device copies per step)
            for it in range(2):
                opt.zero_grad()
                (8 * BCEAndDiceLoss()(m(xs[it]), ys[it])).backward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(10):
                opt.zero_grad()
                loss = BCEAndDiceLoss()(m(xs[it % 4]), ys[it % 4])
                (8 * loss).backward()
                opt.step()
            host[mode] = (time.perf_counter() - t0) / 10 * 1e3          # enqueue time: nothing synchronises in the loop
            torch.cuda.synchronize()
            res[run] = (losses, ev.cpu(), ev2.cpu(), state)
        finally:
            zf._ZFUnetPlan.use_cplan = True
    (l1, e1, f1, s1), (l0, e0, f0, s0), (l0b, _, _, s0b) = res[0], res[1], res[2]
    assert l1 == l0 == l0b, (l1, l0, l0b)
    assert torch.equal(e1, e0) and torch.equal(f1, f0)
    assert float((e1 - f1).abs().max()) > 0          # the second eval input really went through the replayed list
    for k in s0:
        assert torch.equal(s1[k], s0[k]) and torch.equal(s0b[k], s0[k]), k
    print('host enqueue per step: replayed %.2f ms, eager %.2f ms' % (host[True], host[False]))
    assert host[True] < 0.85 * host[False]          # (0.55-0.6 measured; a wide margin: the box may be busy)


def test_eval_matches_train_statistics_path():
    """validate() path (torch_train.py:248-265): no-grad eval forward uses running statistics."""
    m = _model(8, 0.0, 4.0, 'f32')
    x, _ = train_step_ref.synthetic_batch(2, 64, seed=9)
    sd = zf_unet_ref.default_init_state(filters=8, seed=4)
    m.eval()
    with torch.no_grad():
        out = m(x.cuda())
    ref = zf_unet_ref.forward(sd, x, train=False)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=5e-4 * float(ref.abs().max()))


def test_bench_runs_over_rccl_single_rank():
    """The N > 1 launch line of bench.py (torch.distributed.run, backend nccl = RCCL) with ONE rank and the
    collective path forced on: the bucketed gradient all-reduce on the side stream, the loss-sum all-reduce and
    the parameter broadcast all execute on this GPU.  (Multi-rank numerics: tests/test_dist_cpu.py.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEGNB_DP_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29731', os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3',
           '--warmup', '2', '--batch', '4', '--no-cpu-baseline']
    losses = {}
    # the launch lists replayed from C (cut at the gradient-ready hooks) / the eager launcher / the SGD update of each bucket
    # behind its all-reduce (DataParallel.fuse_optimizer)
    for cplan in ('1', '0', '1 fused'):
        out = subprocess.run(cmd + (['--fuse-optimizer'] if 'fused' in cplan else []), env=dict(env, SEGNB_CPLAN=cplan[0]),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        line = [l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1]
        res = json.loads(line)
        assert res['n_gpus'] == 1 and res['value'] > 0 and np.isfinite(res['final_loss'])
        assert res['launch_plan'] == (cplan[0] == '1')
        assert res['optimizer_in_allreduce_epilogue'] == ('fused' in cplan)
        losses[cplan] = res['final_loss']
    # the three launch modes run the same arithmetic in the same order: the same loss, bit for bit
    assert losses['1'] == losses['0'] == losses['1 fused'], losses


def test_bench_executor_model_over_rccl_single_rank():
    """bench.py --model fcdensenet67 with ONE rank and the collective path forced on: the executor's backward hands finished
    gradient ranges to the all-reduce hook at its cuts, eagerly and from the segmented launch lists; same loss either way."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEGNB_DP_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29741', os.path.join(root, 'bench.py'), '--gpus', '1', '--model', 'fcdensenet67',
           '--batch', '2', '--size', '64', '--steps', '4', '--warmup', '4', '--no-cpu-baseline', '--no-kernel-timer']
    losses = {}
    for cplan in ('1', '0'):
        out = subprocess.run(cmd, env=dict(env, SEGNB_CPLAN=cplan), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        res = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1])
        assert res['launch_plan'] == (cplan == '1') and np.isfinite(res['final_loss'])
        losses[cplan] = res['final_loss']
    assert abs(losses['1'] - losses['0']) <= 5e-3          # (Dropout2d draws are the same: one seeded pool per p)


def test_first_gradient_bucket_is_enqueued_before_backward_ends():
    """Data parallel, ONE rank over RCCL (collective path forced on): the all-reduce of the first gradient bucket -- the
    decoder's gradients, final after ~40 % of the backward -- is issued on the communication stream from the backward's
    gradient-ready hook, i.e. BEFORE the compute stream has finished the backward: its start event precedes the
    end-of-backward event.  (Whether the collective then RUNS beside the backward on 8 GPUs is unmeasured: DESIGN section 6.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, json, torch
root = %r
for p in (os.path.join(root, 'segmentation-networks-benchmark_amd'), root):
    sys.path.insert(0, p)
from segnb import dist as sdist, optim
from lib.models.zf_unet import ZF_UNET
from lib.losses import BCEAndDiceLoss
sdist.init_from_env()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = ZF_UNET().to(dev).train()
dp = sdist.DataParallel(m)
assert dp.active and dp.reserved_cus > 0
opt = optim.SGD(m.parameters(), lr=1e-3)
x = torch.randn(8, 3, 224, 224, device=dev)
y = (torch.rand(8, 1, 224, 224, device=dev) > 0.7).long()
gaps = []
for step in range(4):
    if step == 3:
        dp.trace_overlap()
    opt.zero_grad()
    loss = BCEAndDiceLoss()(m(x), y)
    (8 * loss).backward()
    opt.step()
torch.cuda.synchronize()
t = dp.trace
print(json.dumps({'lead_ms': t['first_bucket'].elapsed_time(t['backward_end']), 'loss': float(loss)}))
torch.distributed.destroy_process_group()
""" % root
    env = dict(os.environ, SEGNB_DP_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT='29751', SEGNB_DP_RESERVE_CUS='8')      # (the knob's default is 0)
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    import json
    res = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1])
    print('first bucket enqueued %.3f ms before the end of backward' % res['lead_ms'])
    assert res['lead_ms'] > 0.2 and np.isfinite(res['loss'])


def test_sgd_step_fused_with_the_weight_pack_is_invisible():
    """segnb.optim.SGD.step() through the plan's hook (segnb_sgd_pack_pair_multi + segnb_sgd_ranges: the update of every convolution
    weight inside the pack kernel that writes the NEXT forward's matrices, the other parameters by ranges) == the one-launch
    segnb_sgd_step followed by the pack at the next forward (torch_train.py:71,190: plain SGD): every parameter bit-identical after
    each of three steps, the same losses, and a parameter edited by hand between two steps is packed again (the fused pack's key no
    longer matches)."""
    from lib.losses import BCEAndDiceLoss
    from lib.models.zf_unet import ZF_UNET
    from segnb import optim
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 3, 64, 96, generator=g).cuda()
    y = (torch.rand(2, 1, 64, 96, generator=g) > 0.7).long().cuda()
    keep = optim.SGD.fuse_pack
    out = []
    try:
        for fused in (False, True):
            optim.SGD.fuse_pack = fused
            torch.manual_seed(5)
            m = ZF_UNET(dropout_val=0.0).cuda().train()
            opt = optim.SGD(m.parameters(), lr=0.05)
            losses, snaps = [], []
            for step in range(4):
                if step == 3:
                    with torch.no_grad():          # a hand edit between two steps: the next forward must see it
                        m.conv_224.l1.conv.weight.mul_(0.5)
                opt.zero_grad()
                loss = BCEAndDiceLoss()(m(x), y)
                (2 * loss).backward()
                opt.step()
                torch.cuda.synchronize()
                losses.append(loss.item())
                snaps.append({n: p.detach().clone() for n, p in m.named_parameters()})
            if fused:
                assert m._engine._pairs_key is not None, 'the fused path did not run'
            out.append((losses, snaps))
    finally:
        optim.SGD.fuse_pack = keep
    (l0, s0), (l1, s1) = out
    assert l0 == l1, (l0, l1)
    for a, b in zip(s0, s1):
        for n in a:
            assert torch.equal(a[n], b[n]), n


def test_consumer_side_batchnorm_matches_the_activation_pass():
    """SEGNB_CONSUMER_FUSION: at the 224 x 224 level the second convolution of a block (and its weight gradient) applies the
    first one's BatchNorm + ReLU while it loads (segnb_conv_fprop_tf / segnb_conv_wgrad_tf) -- the activated tensor of
    zf_unet.py:12-17 between them is never written.  Same step as the plan with the activation passes: loss and logits to
    bf16 rounding of a few elements, every gradient in the same direction."""
    from lib.losses import BCEAndDiceLoss
    from lib.models.zf_unet import ZF_UNET
    from segnb import engine
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 224, 224, generator=g).cuda()
    y = (torch.rand(4, 1, 224, 224, generator=g) > 0.7).long().cuda()
    res = []
    keep = engine.Stage.consumer_fusion
    try:
        for fus in (False, True):
            engine.Stage.consumer_fusion = fus
            torch.manual_seed(5)
            m = ZF_UNET(dropout_val=0.0).cuda().train()
            for _ in range(2):                     # (second step: the recorded lists replay)
                m.zero_grad()
                out = m(x)
                loss = BCEAndDiceLoss()(out, y)
                (4 * loss).backward()
            torch.cuda.synchronize()
            res.append((out.detach().float().cpu(), loss.item(), {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters()}))
    finally:
        engine.Stage.consumer_fusion = keep
    (o0, l0, g0), (o1, l1, g1) = res
    assert abs(l0 - l1) < 2e-4, (l0, l1)
    assert float((o0 - o1).abs().max()) < 3e-2 * float(o0.abs().max())
    for n in g0:
        a, b = g0[n].flatten(), g1[n].flatten()
        if float(a.norm()) < 1e-6 * max(float(v.norm()) for v in g0.values()):
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        assert cos > 0.98, (n, cos)


def test_gradient_buffer_cleared_beside_the_forward_and_classifier_fusion_are_invisible():
    """Two host-side schedules that must not change a number (torch_train.py:180-190: zero_grad -> model -> loss -> backward):
    FlatParams.prezero (the flat gradient buffer cleared on the side stream during the forward instead of at the start of
    backward) and the fused last-layer / classifier launches (segnb_bn_fwd_fused_head, segnb_head_bn_bwd).  Three steps with
    zero_grad() between them; and the accumulate-in-place mode (no zero_grad: every .grad aliases the buffer, which must NOT be
    cleared -- lib/train_utils.py:54-65) on top."""
    from lib.losses import BCEAndDiceLoss
    from lib.models.zf_unet import ZF_UNET, _ZFUnetPlan as ZFUnetEngine
    from segnb import engine
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 64, 96, generator=g).cuda()
    y = (torch.rand(4, 1, 64, 96, generator=g) > 0.7).long().cuda()
    keep = engine.FlatParams.prezero_grads, ZFUnetEngine.HEAD_FUSION
    res = {}
    try:
        for mode in ((True, True), (False, True), (True, False)):
            engine.FlatParams.prezero_grads, ZFUnetEngine.HEAD_FUSION = mode
            torch.manual_seed(2)
            m = ZF_UNET(dropout_val=0.0, filters=16).cuda().train()
            out_g = []
            for step in range(3):
                m.zero_grad()
                loss = BCEAndDiceLoss()(m(x), y)
                (4 * loss).backward()
                out_g.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
            # accumulation: no zero_grad -> gradients add up in place
            loss = BCEAndDiceLoss()(m(x), y)
            (4 * loss).backward()
            torch.cuda.synchronize()
            acc = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
            res[mode] = (out_g, acc, float(loss.detach()))
    finally:
        engine.FlatParams.prezero_grads, ZFUnetEngine.HEAD_FUSION = keep
    ref_g, ref_acc, ref_loss = res[(False, True)]
    pre_g, pre_acc, pre_loss = res[(True, True)]
    assert pre_loss == ref_loss
    for step in range(3):
        for n in ref_g[step]:
            assert torch.equal(pre_g[step][n], ref_g[step][n]), (step, n)        # the clear moved, nothing else
    for n in ref_acc:
        assert torch.equal(pre_acc[n], ref_acc[n]), n
        # accumulated = twice the single-step gradient (same weights, same batch, no dropout) to fp32 rounding of the sums
        torch.testing.assert_close(ref_acc[n], 2 * ref_g[2][n], rtol=2e-2, atol=2e-3 * float(ref_g[2][n].abs().max()) + 1e-12)
    # classifier fusion: same dz bit for bit, dw / db / sums in another summation order
    unf_g, _, unf_loss = res[(True, False)]
    assert abs(unf_loss - pre_loss) < 1e-5
    gmax = max(float(v.abs().max()) for v in unf_g[2].values())
    for n in unf_g[2]:
        torch.testing.assert_close(pre_g[2][n], unf_g[2][n], rtol=2e-2, atol=2e-3 * max(float(unf_g[2][n].abs().max()), 1e-3 * gmax))
