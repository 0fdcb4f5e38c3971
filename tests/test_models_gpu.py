"""-m gpu: UNet16 / LinkNet34 / FCDenseNet on the real HIP kernels (fp32 exact path vs oracle / reference golden;
bf16 throughput path: finite, loss within bf16 tolerance, gradient direction sane)."""
import os

import numpy as np
import pytest
import torch

import model_checks as mc

pytestmark = pytest.mark.gpu


def test_tiramisu_f32_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'tiramisu_small.npz'))
    m, _, _, _ = mc.make_tiramisu(g)
    mc.check_tiramisu_golden(m, g, 'cuda')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_fcdensenet57_vs_reference_golden(golden_dir, dtype):
    """FCDenseNet57 (growth rate 12, tiramisu.py:187-191): padded slices + per-slice BatchNorm on the HIP kernels"""
    g = np.load(os.path.join(golden_dir, 'tiramisu57_small.npz'))
    m = mc.make_tiramisu57_golden(g)
    print('fcdensenet57 %s dloss %.2e diou %.2e' % ((dtype,) + mc.check_product_golden(m, g, 'cuda', dtype)))


def test_executor_reduce_in_the_consumers_data_gradient():
    mc.check_executor_fused_reduce('cuda')


def test_linknet34_abs_eps_affine_form_inside_the_plan():
    """lib/modules/abn/functions.py:94-118 form (scale |w| + eps, signed weight gradient) inside LinkNet34's fused plan"""
    mc.check_linknet_abs_eps_form('cuda')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_unet16_vs_oracle(dtype):
    m, fwd, x, y = mc.make_unet16()
    print('unet16 %s cosine %.6f' % (dtype, mc.check_against_oracle(m, fwd, x, y, 'cuda', dtype)))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_linknet34_vs_oracle(dtype):
    m, fwd, x, y = mc.make_linknet(128)
    print('linknet34 %s cosine %.6f' % (dtype, mc.check_against_oracle(m, fwd, x, y, 'cuda', dtype, min_cos=0.9999)))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_tiramisu_vs_oracle(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, 'tiramisu_small.npz'))
    m, fwd, x, y = mc.make_tiramisu(g)
    print('tiramisu %s cosine %.6f' % (dtype, mc.check_against_oracle(m, fwd, x, y, 'cuda', dtype)))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_unet16_vs_reference_golden(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, 'unet16_small.npz'))
    m, _ = mc.make_unet16_golden(g)
    print('unet16 %s dloss %.2e diou %.2e' % ((dtype,) + mc.check_product_golden(m, g, 'cuda', dtype)))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_unet16_bilinear_decoder_vs_reference_golden(golden_dir, dtype):
    """UNet16(is_deconv=False): DecoderBlock's bilinear branch (unet16.py:42-46) on the HIP path vs the fixture the reference's own
    DecoderBlock(is_deconv=False) produced."""
    g = np.load(os.path.join(golden_dir, 'unet16_bilinear_small.npz'))
    m, _ = mc.make_unet16_bilinear_golden(g)
    print('unet16 bilinear %s dloss %.2e diou %.2e' % ((dtype,) + mc.check_product_golden(m, g, 'cuda', dtype)))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_linknet34_vs_reference_golden(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, 'linknet_small.npz'))
    m, _ = mc.make_linknet_golden(g)
    print('linknet34 %s dloss %.2e diou %.2e' % ((dtype,) + mc.check_product_golden(m, g, 'cuda', dtype)))


def test_training_steps_reduce_loss_all_models():
    """torch_train.py:180-190 loop on each model family, bf16, a few SGD steps on a fixed batch."""
    import warnings
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.linknet import LinkNet34
    from lib.models.tiramisu import FCDenseNet67
    from lib.models.unet16 import UNet16
    from segnb import optim
    torch.manual_seed(0)
    x = torch.randn(4, 3, 96, 96).cuda()
    y = (torch.rand(4, 1, 96, 96) > 0.7).long().cuda()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        models = [UNet16(), LinkNet34(), FCDenseNet67(n_classes=1)]
    for m in models:
        m = m.cuda().train()
        opt = optim.SGD(m.parameters(), lr=1e-2)
        crit = BCEWithLogitsLossAndSmoothJaccard()
        losses = []
        for _ in range(8):
            opt.zero_grad()
            loss = crit(m(x), y)
            (x.size(0) * loss).backward()
            opt.step()
            losses.append(loss.item())
        assert all(np.isfinite(losses)), (type(m).__name__, losses)
        assert losses[-1] < losses[0], (type(m).__name__, losses)


@pytest.mark.parametrize('which', ['unet16', 'zf_unet'])
def test_stored_weight_gradients_accumulate_when_the_caller_does(which):
    """Directly delivered weight gradients (segnb_wgrad_target) are STORED when the backward cleared the flat gradient buffer
    (zero_grad() before every step, torch_train.py:180) and ADDED when gradients accumulate in place (lib/train_utils.py:54-65 never
    zeroes them).  A recorded launch list carries the mode it was recorded under: three zeroed steps (eager, recorded, replayed), then
    two backwards WITHOUT zero_grad on the same batch and weights -- the gradient must be exactly twice, then three times, the
    single one (no dropout, no optimizer step) to fp32 rounding of the sums, whichever list serves the step."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    crit = BCEWithLogitsLossAndSmoothJaccard()
    if which == 'unet16':
        m, _, x, y = mc.make_unet16()
    else:
        from lib.models.zf_unet import ZF_UNET
        torch.manual_seed(2)
        m = ZF_UNET(dropout_val=0.0, filters=8)
        g = torch.Generator().manual_seed(4)
        x, y = torch.randn(2, 3, 64, 96, generator=g), (torch.rand(2, 1, 64, 96, generator=g) > 0.7).long()
    m = m.cuda().train()
    x, y = x.cuda(), y.cuda()
    for mod in m.modules():                       # (BatchNorm running statistics move with every forward: they do not enter the loss)
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0

    def backward():
        loss = crit(m(x), y)
        (x.size(0) * loss).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    for _ in range(3):
        m.zero_grad()
        single = backward()
    twice = backward()                            # no zero_grad: every .grad aliases the flat buffer and is accumulated into
    thrice = backward()
    worst = 0.0
    for n in single:
        s = float(single[n].abs().max())
        if s == 0.0:
            assert float(twice[n].abs().max()) == 0.0
            continue
        worst = max(worst, float((twice[n] - 2 * single[n]).abs().max()) / s, float((thrice[n] - 3 * single[n]).abs().max()) / s)
    assert worst < 1e-4, worst
    m.zero_grad()                                 # and back: a zeroed step after the accumulating ones
    again = backward()
    for n in single:
        torch.testing.assert_close(again[n], single[n], rtol=1e-5, atol=1e-6 * float(single[n].abs().max() + 1e-30))


def test_weight_gradient_cu_share_is_local_to_the_model():
    """UNet16 takes ALL CUs for its weight gradients (HipNet.wg_cu_pct = 100 through segnb_wg_cu_share: its side stream is the
    longer one, unet16.py:50-108), LinkNet34 in the same process keeps the library's default half: the share is set around the
    planning and the launches of one model's convolutions -- eager and replayed -- and restored behind them.  Interleaved
    training steps of both models (the second and third of each replay recorded launch lists) must leave the default in place
    and give the gradients the models give alone."""
    import warnings
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.linknet import LinkNet34
    from lib.models.unet16 import UNet16
    from segnb import _native as nv
    from segnb import convplan as cp
    from segnb.engine import ConvOp, Runtime
    assert UNet16.wg_cu_pct == 100 and LinkNet34.wg_cu_pct is None
    rt = Runtime('cuda', 'bf16')
    probe = ConvOp(rt, torch.randn(128, 128, 3, 3).cuda(), None, [(128, 128)], 1, 1, False, True)
    default = list(probe.plan(64, 64)['nslab'])
    probe2 = ConvOp(rt, torch.randn(128, 128, 3, 3).cuda(), None, [(128, 128)], 1, 1, False, True)
    probe2.wg_cu_pct = 100
    if Runtime.overlap_wgrad:           # (SEGNB_OVERLAP_WGRAD=0: one stream, the default share IS all of the CUs)
        assert probe2.plan(64, 64)['nslab'][0] > default[0]          # (a full share is more pixel splits)
    x = torch.randn(2, 3, 128, 128).cuda()
    y = (torch.rand(2, 1, 128, 128) > 0.7).long().cuda()
    crit = BCEWithLogitsLossAndSmoothJaccard()

    def step(m):
        torch.manual_seed(7)                 # (LinkNet34 draws Dropout2d masks: the same ones every step)
        m.zero_grad()
        loss = crit(m(x), y)
        (2 * loss).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(1)
        u, l = UNet16().cuda().train(), LinkNet34().cuda().train()
    alone = {}
    for name, m in (('u', u), ('l', l)):
        for _ in range(3):
            alone[name] = step(m)
    for it in range(3):
        gu = step(u)
        fresh = ConvOp(rt, torch.randn(128, 128, 3, 3).cuda(), None, [(128, 128)], 1, 1, False, True)
        assert list(fresh.plan(64, 64)['nslab']) == default, 'UNet16 left its share behind'
        gl = step(l)
    for got, ref in ((gu, alone['u']), (gl, alone['l'])):
        scale = max(float(v.norm()) for v in ref.values())
        for k in ref:
            if float(ref[k].norm()) < 1e-6 * scale:      # (a bias in front of a BatchNorm: an exactly zero gradient, noise on both sides)
                continue
            # (not bitwise: the BatchNorm sums are fp64 atomics and a flipped bf16 rounding travels down the net; a weight gradient
            # computed under the wrong share -- slabs of another count -- would be off by whole slabs, not by rounding)
            err = float((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-20))
            assert err < 2e-2, (k, err)


@pytest.mark.parametrize('name', ['unet16', 'linknet34', 'fcdensenet67'])
def test_launch_plans_replay_matches_eager(name):
    """segnb_plan_*: from the third step of a geometry on, the executor-driven models replay their recorded forward and
    backward launch lists from C.  Same seeds with the replay on and off: the losses of seven SGD steps agree (Dropout2d
    on: the pools are redrawn outside the lists), an eval forward of another batch agrees, and the lists are in use."""
    import warnings
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.linknet import LinkNet34
    from lib.models.tiramisu import FCDenseNet67
    from lib.models.unet16 import UNet16
    from segnb import net, optim
    make = {'unet16': UNet16, 'linknet34': LinkNet34, 'fcdensenet67': lambda: FCDenseNet67(n_classes=1)}[name]
    gen = torch.Generator().manual_seed(11)
    xs = [torch.randn(4, 3, 64, 64, generator=gen).cuda() for _ in range(3)]
    ys = [(torch.rand(4, 1, 64, 64, generator=gen) > 0.7).long().cuda() for _ in range(3)]
    res = {}
    for mode in (True, False):
        net.HipNet.use_cplan = mode
        try:
            torch.manual_seed(0)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                m = make().set_compute_dtype('f32').cuda().train()
            opt = optim.SGD(m.parameters(), lr=1e-3)
            crit = BCEWithLogitsLossAndSmoothJaccard()
            losses = []
            for it in range(7):
                opt.zero_grad()
                loss = crit(m(xs[it % 3]), ys[it % 3])
                (4 * loss).backward()
                if it == 6:                # (a replayed step: EVERY gradient, incl. the batched bias gradients launched from the host)
                    last_grads = {k: p.grad.detach().clone().cpu() for k, p in m.named_parameters() if p.grad is not None}
                opt.step()
                losses.append(loss.item())
            m.eval()
            with torch.no_grad():
                evs = [m(xs[i]).clone().cpu() for i in (0, 1, 2, 1)]
            plans = m._tape.plans
            if mode:
                ready = [e for e in plans.values() if e.get('state') == 'ready']
                assert len(ready) >= 2, [e.get('state') for e in plans.values()]          # train fwd+bwd, eval fwd
                assert any(e.get('nbwd', 0) > 10 for e in ready) and all(e['nfwd'] > 10 for e in ready)
            else:
                assert not plans
            res[mode] = (losses, evs, last_grads)
        finally:
            net.HipNet.use_cplan = True
    (l1, e1, g1), (l0, e0, g0) = res[True], res[False]
    scale = max(float(v.norm()) for v in g0.values())
    for k in g0:
        err = float((g1[k] - g0[k]).norm())
        # (fp32 parity mode: the general weight-gradient kernel's float atomics and ReLUs within an ulp of zero make two runs of
        # the SAME launcher differ by this much on the smallest tensors; a gradient missing from the replay is off by its norm)
        assert err <= 5e-2 * float(g0[k].norm()) + 1e-5 * scale, (k, err, float(g0[k].norm()), float(g1[k].norm()))
    np.testing.assert_allclose(l1, l0, rtol=0, atol=2e-3)
    for a, b in zip(e1, e0):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max())
    assert float((e1[3] - e1[1]).abs().max()) == 0.0 and float((e1[0] - e1[1]).abs().max()) > 0


# ---- the remaining BASELINE.json configurations at their own sizes (configs_untested) ------------------
def _full_size_bf16_vs_fp32(make, B, S, tag):
    """One training step (torch_train.py:180-190 body, bce_jaccard) of `make()` at the configuration's size on the
    bf16 throughput path vs the exact-fp32 HIP path (the one pinned to the reference fixtures at small sizes), same
    weights / batch, Dropout2d off; plus size-independent properties: finite, run-to-run reproducible eval forward,
    loss goes down under SGD."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore
    from segnb import optim
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, S, S, generator=gen).cuda()
    y = (torch.rand(B, 1, S, S, generator=gen) > 0.7).long().cuda()
    res = {}
    for dtype in ('f32', 'bf16'):
        torch.manual_seed(0)
        m = make().set_compute_dtype(dtype).cuda().train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        out = m(x)
        loss = BCEWithLogitsLossAndSmoothJaccard()(out, y)
        iou = JaccardScore()(out, y).item()
        (B * loss).backward()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        g = torch.cat([p.grad.detach().double().reshape(-1).cpu() for p in m.parameters() if p.dim() == 4])
        res[dtype] = (loss.item(), iou, g, out.detach().cpu())
        if dtype == 'bf16':
            m.eval()
            with torch.no_grad():
                a, b = m(x).clone(), m(x).clone()
            assert torch.equal(a, b), 'eval forward is not run-to-run reproducible'
            m.train()
            opt = optim.SGD(m.parameters(), lr=1e-3)
            losses = []
            for _ in range(4):
                opt.zero_grad()
                l = BCEWithLogitsLossAndSmoothJaccard()(m(x), y)
                (B * l).backward()
                opt.step()
                losses.append(l.item())
            assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        del m
        torch.cuda.empty_cache()
    (l32, i32, g32, o32), (l16, i16, g16, o16) = res['f32'], res['bf16']
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    print('%s bs=%d %dx%d bf16 - fp32: dloss %.3e dIoU %.3e, conv-weight gradient cosine %.4f, norm ratio %.4f'
          % (tag, B, S, S, l16 - l32, i16 - i32, cos, float(g16.norm() / g32.norm())))
    assert abs(l16 - l32) < 5e-3 and abs(i16 - i32) < 5e-3
    assert cos > 0.7 and abs(float(g16.norm() / g32.norm()) - 1.0) < 0.15, cos


def test_fcdensenet103_256_bs8_config():
    """BASELINE.json configs[3]: FCDenseNet-103 256x256 bs=8 (per-GPU shard of the data-parallel job)."""
    from lib.models.tiramisu import FCDenseNet103
    _full_size_bf16_vs_fp32(lambda: FCDenseNet103(n_classes=1), 8, 256, 'FCDenseNet103')


def test_linknet34_512_bs16_config():
    """BASELINE.json configs[2]: LinkNet-34 512x512 bf16 bs=16."""
    import warnings
    from lib.models.linknet import LinkNet34

    def make():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return LinkNet34()
    _full_size_bf16_vs_fp32(make, 16, 512, 'LinkNet34')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_replay_fcdensenet103_structure(dtype):
    """Teacher-forced replay (tests/abi_replay.py) of the REAL FCDenseNet103 (all 103 layers, growth 16, 1072-channel
    dense inputs) at a small spatial size: every launch of the plan vs the emulator at per-op tolerance."""
    import abi_replay
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.tiramisu import FCDenseNet103
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 64, 64, generator=gen)
    y = (torch.rand(2, 1, 64, 64, generator=gen) > 0.7).long()

    def make():
        torch.manual_seed(3)
        m = FCDenseNet103(n_classes=1)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        return m
    n, rep = abi_replay.replay(make, x, y, BCEWithLogitsLossAndSmoothJaccard(), dtype)
    assert n > 900 and not rep, '%d of %d calls differ:\n%s' % (len(rep), n, '\n'.join(rep[:20]))


def test_unet16_1024_tiled_config():
    """BASELINE.json configs[4]: default UNet16 (TernausNet VGG16-UNet, 32.2 M parameters) on 1024x1024 tiles, bs=4,
    through lib/tiles.py sliding-window inference with D4 TTA (segnb.tiled.predict_tiled).
    (1) the eval forward of one 1024x1024 tile, fp32 HIP path and bf16 path, vs the oracle restatement
        (oracle/unet16_ref.py, pinned to the reference's unet16.py by unet16_small.npz) on the CPU;
    (2) predict_tiled over a 1100x1300 image (4 tiles x 8 transforms, last batch short) == the oracle's restatement
        of inria_submit.predict_tiled fed by the same model."""
    from lib.models.unet16 import UNet16
    from oracle import tiles_ref, unet16_ref
    from segnb.tiled import predict_tiled
    torch.manual_seed(0)
    m = UNet16().cuda().eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(2)
    x1 = torch.randn(1, 3, 1024, 1024, generator=gen)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = unet16_ref.forward(sd, x1)
    scale = float(ref.abs().max())
    for dtype, tol in (('f32', 5e-4), ('bf16', 6e-2)):
        m.set_compute_dtype(dtype)
        with torch.no_grad():
            got = m(x1.cuda()).cpu()
        err = float((got - ref).abs().max())
        print('UNet16 1024x1024 eval forward %s vs oracle: max|d| %.3e of scale %.3e' % (dtype, err, scale))
        assert err <= tol * scale, (dtype, err, scale)
    rng = np.random.RandomState(4)
    img = rng.randn(1100, 1300, 3).astype(np.float32)
    got = predict_tiled(img, m, None, 1024, 4)

    def logits_fn(xb):
        with torch.no_grad():
            return m(torch.from_numpy(xb).cuda()).float().cpu().numpy()
    ref = tiles_ref.predict_tiled(img, logits_fn, 1024, 4)
    assert got.shape == (1100, 1300, 1) and np.isfinite(got).all()
    np.testing.assert_allclose(got[..., 0], ref[..., 0] if ref.ndim == 3 else ref, rtol=1e-5, atol=1e-5)
    # (3) the uint8 image with a NormalizeImage transform (inria_submit.py:286-288), normalised inside the gather kernel
    #     (segnb_tiles_gather_u8: what `bench.py --tiled` times on a 5000 x 5000 image) == the oracle fed the host-normalised image
    from segnb.engine import InputNorm
    img8 = rng.randint(0, 256, size=(1100, 1300, 3), dtype=np.uint8)
    nm = InputNorm(mean=(0.40, 0.42, 0.38), std=(0.19, 0.18, 0.18))
    timing = {}
    got8 = predict_tiled(img8, m, nm, 1024, 4, timing=timing)
    host = ((img8 * nm.scale - np.array(nm.mean, dtype=np.float32)) / np.array(nm.std, dtype=np.float32)).astype(np.float32)
    ref8 = tiles_ref.predict_tiled(host, logits_fn, 1024, 4)
    np.testing.assert_allclose(got8[..., 0], ref8[..., 0] if ref8.ndim == 3 else ref8, rtol=1e-4, atol=2e-4)
    assert timing['nitems'] == 32 and set(timing['ms']) >= {'upload', 'gather', 'forward', 'merge', 'download'}


def test_inplace_abn_standalone_gpu():
    """lib.modules.abn.InPlaceABN called on its own (bn.py:47-103) on the HIP kernels == BatchNorm2d + LeakyReLU(0.01)
    on the CPU, forward / backward / running statistics; eval mode too."""
    from lib.modules.abn import InPlaceABN
    torch.manual_seed(3)
    x = torch.randn(3, 12, 9, 7)
    abn = InPlaceABN(12).cuda()
    bn = torch.nn.BatchNorm2d(12)
    with torch.no_grad():
        abn.weight.copy_(1 + 0.2 * torch.randn(12))
        abn.bias.copy_(0.1 * torch.randn(12))
        bn.weight.copy_(abn.weight.cpu())
        bn.bias.copy_(abn.bias.cpu())
    xa, xb = x.clone().cuda().requires_grad_(True), x.clone().requires_grad_(True)
    ya = abn(xa)
    yb = torch.nn.functional.leaky_relu(bn(xb), 0.01)
    r = torch.randn_like(yb)
    (ya * r.cuda()).sum().backward()
    (yb * r).sum().backward()
    torch.testing.assert_close(ya.cpu(), yb, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xa.grad.cpu(), xb.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(abn.weight.grad.cpu(), bn.weight.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(abn.bias.grad.cpu(), bn.bias.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(abn.running_mean.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(abn.running_var.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    # in place, like the reference (functions.py:92): a non-leaf input receives the result in its own storage
    bn2 = torch.nn.BatchNorm2d(12)
    bn2.load_state_dict(bn.state_dict())
    abn2 = InPlaceABN(12).cuda()
    abn2.load_state_dict({k: v for k, v in bn.state_dict().items() if k != 'num_batches_tracked'})
    xc, xd = x.clone().cuda().requires_grad_(True), x.clone().requires_grad_(True)
    h = xc * 2.0
    yc = abn2(h)
    assert yc.data_ptr() == h.data_ptr()
    yd = torch.nn.functional.leaky_relu(bn2(xd * 2.0), 0.01)
    (yc * r.cuda()).sum().backward()
    (yd * r).sum().backward()
    torch.testing.assert_close(yc.cpu(), yd, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xc.grad.cpu(), xd.grad, rtol=1e-4, atol=1e-5)
    abn.eval()
    bn.eval()
    with torch.no_grad():
        torch.testing.assert_close(abn(x.cuda()).cpu(), torch.nn.functional.leaky_relu(bn(x), 0.01), rtol=1e-5, atol=1e-5)


def test_inplace_abn_constructor_surface_gpu():
    mc.check_inplace_abn_surface('cuda')


def test_inplace_abn_abs_eps_affine_form_gpu():
    mc.check_inplace_abn_abs_form('cuda')


def test_find_optimal_lr_gpu():
    """lib.train_utils.find_optimal_lr (train_utils.py:36-69) on the HIP path: 30 steps, lr doubling from 1e-8,
    gradients accumulate (never zeroed) exactly as the same loop on the oracle."""
    from lib.losses import BCEWithSigmoidLoss
    from lib.models.zf_unet import ZF_UNET
    from lib.train_utils import find_optimal_lr
    from oracle import train_step_ref, zf_unet_ref
    B, S, F = 2, 64, 8
    x, y = train_step_ref.synthetic_batch(B, S, seed=13)
    torch.manual_seed(5)
    m = ZF_UNET(dropout_val=0.0, filters=F).set_compute_dtype('f32').cuda()
    opt = torch.optim.SGD(m.parameters(), lr=1.0)
    lrs, loss = find_optimal_lr(m, BCEWithSigmoidLoss(), opt, [(x, y)] * 30)
    assert lrs.shape == (30,) and abs(lrs[0] - 1e-8) < 1e-12 and np.all(np.isfinite(loss))
    # the same 30-step loop on the oracle (accumulating gradients, lr table through the base lr 1.0)
    sd = zf_unet_ref.default_init_state(filters=F, seed=5)
    acc = None
    ref = []
    for i in range(30):
        l, _, grads = train_step_ref.loss_and_grads(sd, x, y, 'bce')
        acc = grads if acc is None else {k: acc[k] + grads[k] for k in grads}
        for k in acc:
            sd[k] = sd[k] - float(lrs[i]) * acc[k]
        ref.append(l.item())
    ref = np.array(ref, dtype=np.float32)
    # the first half of the schedule (lr <= 1e-4) is a fixed point up to fp32 noise; once the steps become large the
    # trajectory amplifies that noise (random-init net, batch of 2): bounded relative to the loss
    np.testing.assert_allclose(loss[:14], ref[:14], rtol=0, atol=2e-5)
    np.testing.assert_allclose(loss[14:], ref[14:], rtol=2e-2, atol=0)
