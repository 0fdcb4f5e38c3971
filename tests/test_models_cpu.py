"""UNet16 / LinkNet34 / FCDenseNet plans (segnb.net executor + lib.models.*) on CPU through the ABI emulator:
pins the host logic of the three model families (geometry of stride-2 / transposed / 7x7 / 2x2 convs, residual
and skip-add gradient routing, dense-block buffers, center_crop, InPlaceABN holders, state_dict layouts)."""
import os

import numpy as np
import pytest
import torch

import model_checks as mc
from oracle import abi_emulator, tiramisu_ref, losses_ref
from segnb import _native as nv


@pytest.fixture(autouse=True)
def emulated_abi():
    nv.set_backend_for_testing(abi_emulator.AbiEmulator())
    yield
    nv.set_backend_for_testing(None)


def test_tiramisu_oracle_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'tiramisu_small.npz'))
    sd = {k[3:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith('sd/')}
    leaves = {k: (v.requires_grad_(True) if ('running' not in k and v.is_floating_point()) else v) for k, v in sd.items()}
    cfg = mc.TIRAMISU_CFG
    out = tiramisu_ref.forward(leaves, torch.from_numpy(g['x']), cfg['down_blocks'], cfg['up_blocks'],
                               cfg['bottleneck_layers'], True)
    np.testing.assert_allclose(out.detach().numpy(), g['train_logits'], rtol=1e-5, atol=1e-6)
    loss = losses_ref.bce_jaccard(out, torch.from_numpy(g['y']))
    assert abs(loss.item() - float(g['loss_bce_jaccard'])) < 1e-6
    (2 * loss).backward()
    gmax = max(np.abs(g['grad/' + k]).max() for k in leaves if leaves[k].requires_grad)
    for k, v in leaves.items():
        if v.requires_grad and np.abs(g['grad/' + k]).max() > 1e-6 * gmax:
            np.testing.assert_allclose(v.grad.numpy(), g['grad/' + k], rtol=1e-3, atol=1e-6 * gmax, err_msg=k)


def test_tiramisu_product_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'tiramisu_small.npz'))
    m, _, _, _ = mc.make_tiramisu(g)
    mc.check_tiramisu_golden(m, g, 'cpu')


def test_tiramisu_with_never_stored_data_gradients(monkeypatch):
    """The optional two-launch form of the dense layers' data gradients (Tape.two_launch_dgrad: segnb_conv_fprop_bnsums / _bnapply,
    off by default) gives the gradients of the stored form, and is actually taken (prefixes of more than 32 channels)."""
    from lib.models.tiramisu import FCDenseNet
    from lib.losses import BCEWithSigmoidLoss
    from segnb import net as NN
    cfg = dict(in_channels=3, down_blocks=(2, 2), up_blocks=(2, 2), bottleneck_layers=2, growth_rate=16, out_chans_first_conv=48,
               n_classes=1)
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    y = (torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(6)) > 0.5).long()

    def run(flag):
        monkeypatch.setattr(NN.Tape, 'two_launch_dgrad', flag)
        torch.manual_seed(3)
        m = FCDenseNet(**cfg)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        m.set_compute_dtype('bf16').train()          # (the two launches are bf16 kernels)
        seen = []
        orig = nv.call

        def call(name, *a):
            seen.append(name)
            return orig(name, *a)
        monkeypatch.setattr(nv, 'call', call)
        (2 * BCEWithSigmoidLoss()(m(x), y)).backward()
        monkeypatch.setattr(nv, 'call', orig)
        return {n: p.grad.clone() for n, p in m.named_parameters()}, seen

    g0, s0 = run(False)
    g1, s1 = run(True)
    assert s0.count('segnb_conv_fprop_bnsums') == 0
    assert s1.count('segnb_conv_fprop_bnsums') >= 4 and s1.count('segnb_conv_fprop_bnsums') == s1.count('segnb_conv_fprop_bnapply')
    for n in g0:
        scale = max(float(g0[n].abs().max()), 1e-6)
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-5 * scale, n      # (the emulator composes both forms from the same passes)


def test_fcdensenet57_product_vs_reference_golden(golden_dir):
    """FCDenseNet57(n_classes) of tiramisu.py:187-191 -- growth rate 12, every slice of the concat buffers padded to 16
    channels, BatchNorm per slice -- against the fixture the reference's own FCDenseNet57 produced (make_golden.py)"""
    g = np.load(os.path.join(golden_dir, 'tiramisu57_small.npz'))
    m = mc.make_tiramisu57_golden(g)
    assert sum(p.numel() for p in m.parameters()) == 1374865
    mc.check_product_golden(m, g, 'cpu')


def test_fcdensenet_any_growth_rate_vs_reference_module():
    """growth rates / first-convolution widths that are not multiples of 8 (tiramisu.py:94-96 takes any) against the
    state_dict-compatible torch restatement in oracle/tiramisu_ref.py"""
    from lib.models.tiramisu import FCDenseNet
    cfg = dict(in_channels=3, down_blocks=(1, 3), up_blocks=(3, 1), bottleneck_layers=1, growth_rate=10, out_chans_first_conv=12,
               n_classes=2)
    torch.manual_seed(3)
    m = FCDenseNet(**cfg)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    m.set_compute_dtype('f32').train()
    x = torch.randn(2, 3, 20, 24)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = tiramisu_ref.forward(sd, x, cfg['down_blocks'], cfg['up_blocks'], cfg['bottleneck_layers'], True)
    out = m(x)
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)


def test_tiramisu_state_dict_and_factories():
    from lib.models.tiramisu import FCDenseNet57, FCDenseNet67, FCDenseNet103
    assert FCDenseNet57(n_classes=3).num_classes == 3
    m = FCDenseNet103(n_classes=1)
    assert sum(p.numel() for p in m.parameters()) == 9319521      # = the reference's FCDenseNet103(1) (probe; SURVEY 8a a4: ~9.3 M)
    assert 'denseBlocksDown.0.layers.0.norm.weight' in m.state_dict()
    assert 'transUpBlocks.4.convTrans.weight' in m.state_dict()
    assert FCDenseNet67(n_classes=1).num_classes == 1


def test_unet16_vs_oracle():
    m, fwd, x, y = mc.make_unet16()
    assert 'encoder.28.weight' in m.state_dict() and 'conv5.4.weight' in m.state_dict()
    assert 'center.block.1.weight' in m.state_dict() and 'dec1.conv.bias' in m.state_dict()
    mc.check_against_oracle(m, fwd, x, y, 'cpu')


def test_unet16_oracle_and_product_vs_reference_golden(golden_dir):
    """PIN of oracle/unet16_ref.py: the fixture was produced by the reference's unet16.py:52-131."""
    from oracle import unet16_ref
    g = np.load(os.path.join(golden_dir, 'unet16_small.npz'))
    m, fwd = mc.make_unet16_golden(g)
    mc.check_oracle_golden(fwd, fwd, m, g)
    mc.check_product_golden(m, g, 'cpu')


def test_unet16_bilinear_decoder_oracle_and_product_vs_reference_golden(golden_dir):
    """DecoderBlock's other branch (unet16.py:42-46: Upsample(x2, bilinear) -> ConvRelu -> ConvRelu).  The fixture was produced by
    the reference's UNet16 with its decoder blocks replaced by the reference's own DecoderBlock(is_deconv=False) instances
    (make_golden.py gen_unet16_bilinear); pins the oracle's branch and the product's UNet16(is_deconv=False) on the emulator."""
    g = np.load(os.path.join(golden_dir, 'unet16_bilinear_small.npz'))
    m, fwd = mc.make_unet16_bilinear_golden(g)
    assert 'center.block.1.conv.weight' in m.state_dict() and 'dec2.block.2.conv.bias' in m.state_dict()
    mc.check_oracle_golden(fwd, fwd, m, g)
    mc.check_product_golden(m, g, 'cpu')


def test_linknet34_oracle_and_product_vs_reference_golden(golden_dir):
    """PIN of oracle/linknet_ref.py: the fixture was produced by the reference's linknet.py:5-90."""
    from oracle import linknet_ref
    g = np.load(os.path.join(golden_dir, 'linknet_small.npz'))
    m, _ = mc.make_linknet_golden(g)
    mc.check_oracle_golden(lambda sd, x: linknet_ref.forward(sd, x, True), lambda sd, x: linknet_ref.forward(sd, x, False),
                           m, g)
    mc.check_product_golden(m, g, 'cpu')


def test_two_input_sizes_without_a_parameter_update():
    """Packed weights are per input size (ConvOp.plan): a second size with NO optimizer step in between must pack its
    own plan, and going back to the first size after a step must see the new weights (net.py Tape.begin)."""
    from lib.models.unet16 import UNet16
    torch.manual_seed(0)
    m = UNet16(num_filters=4).set_compute_dtype('f32').eval()
    g = torch.Generator().manual_seed(1)
    xa, xb = torch.randn(1, 3, 32, 32, generator=g), torch.randn(1, 3, 64, 32, generator=g)
    with torch.no_grad():
        ya1 = m(xa)
        yb1 = m(xb)
    fresh = UNet16(num_filters=4).set_compute_dtype('f32').eval()
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        torch.testing.assert_close(yb1, fresh(xb), rtol=0, atol=0)
        torch.testing.assert_close(ya1, fresh(xa), rtol=0, atol=0)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.5)
    fresh2 = UNet16(num_filters=4).set_compute_dtype('f32').eval()
    fresh2.load_state_dict(m.state_dict())
    with torch.no_grad():
        torch.testing.assert_close(m(xb), fresh2(xb), rtol=0, atol=0)
        torch.testing.assert_close(m(xa), fresh2(xa), rtol=0, atol=0)     # A's plan was packed before the update


def test_unet16_default_parameter_count():
    from lib.models.unet16 import UNet16
    assert sum(p.numel() for p in UNet16().parameters()) == 32202337


def test_linknet34_vs_oracle():
    m, fwd, x, y = mc.make_linknet()
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 21794721
    for k in ('firstconv.weight', 'firstbn.num_batches_tracked', 'encoder2.0.downsample.0.weight',
              'encoder4.2.bn2.running_var', 'decoder4.abn1.running_mean', 'decoder1.deconv2.weight',
              'finalconv3.bias'):
        assert k in sd, k
    assert 'decoder4.abn1.num_batches_tracked' not in sd               # InPlaceABN has no such buffer (bn.py:77-78)
    mc.check_against_oracle(m, fwd, x, y, 'cpu', min_cos=0.9999)


def test_inplace_abn_module_standalone():
    """lib.modules.abn.InPlaceABN called on its own == BatchNorm2d(train) + LeakyReLU(0.01), fwd and bwd."""
    from lib.modules.abn import InPlaceABN
    torch.manual_seed(3)
    x = torch.randn(3, 12, 9, 7)
    abn = InPlaceABN(12)
    bn = torch.nn.BatchNorm2d(12)
    with torch.no_grad():
        abn.weight.copy_(1 + 0.2 * torch.randn(12)); abn.bias.copy_(0.1 * torch.randn(12))
        bn.weight.copy_(abn.weight); bn.bias.copy_(abn.bias)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = abn(xa)
    yb = torch.nn.functional.leaky_relu(bn(xb), 0.01)
    r = torch.randn_like(ya)
    (ya * r).sum().backward()
    (yb * r).sum().backward()
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(abn.weight.grad, bn.weight.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(abn.bias.grad, bn.bias.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(abn.running_var, bn.running_var, rtol=1e-5, atol=1e-6)
    # in place, like the reference (functions.py:92 mark_dirty): a non-leaf input receives the result in its own storage
    xc, xd = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    h = xc * 2.0
    yc = abn(h)
    assert yc.data_ptr() == h.data_ptr()
    yd = torch.nn.functional.leaky_relu(bn(xd * 2.0), 0.01)
    (yc * r).sum().backward()
    (yd * r).sum().backward()
    torch.testing.assert_close(yc, yd, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xc.grad, xd.grad, rtol=1e-4, atol=1e-5)


def test_inplace_abn_constructor_surface():
    mc.check_inplace_abn_surface('cpu')


def test_executor_reduce_in_the_consumers_data_gradient():
    mc.check_executor_fused_reduce('cpu')


def test_linknet34_abs_eps_affine_form_inside_the_plan():
    mc.check_linknet_abs_eps_form('cpu')


def test_inplace_abn_abs_eps_affine_form():
    mc.check_inplace_abn_abs_form('cpu')


def test_replay_harness_self_consistent(golden_dir):
    """tests/abi_replay.py (the teacher-forced GPU parity harness) replaying the emulator against itself."""
    import abi_replay
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    g = np.load(os.path.join(golden_dir, 'tiramisu_small.npz'))
    _, _, x, y = mc.make_tiramisu(g)
    n, rep = abi_replay.replay(lambda: mc.make_tiramisu(g)[0], x, y, BCEWithLogitsLossAndSmoothJaccard(), 'f32',
                               device='cpu')
    assert n > 150 and not rep, rep[:5]          # (185 ABI calls: the bias gradients are one batched launch)


def test_frozen_parameters_keep_grad_none():
    """requires_grad=False parameters (torch_train_ab.py:245-246 freezes a whole head; a fine-tuning script freezes the
    encoder): their .grad stays None after backward, the trainable ones get the same gradients as in the unfrozen run, and an
    optimizer over the trainable subset updates only those."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(2)
    x = torch.randn(2, 3, 32, 32)
    y = (torch.rand(2, 1, 32, 32) > 0.6).long()
    grads = []
    for freeze in (False, True):
        torch.manual_seed(4)
        m = ZF_UNET(dropout_val=0.0, filters=8).set_compute_dtype('f32').train()
        if freeze:
            for n, p in m.named_parameters():
                if n.startswith('conv_'):
                    p.requires_grad = False
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.1)
        loss = BCEWithLogitsLossAndSmoothJaccard()(m(x), y)
        (2 * loss).backward()
        grads.append({n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()})
        opt.step()
        if freeze:
            for n, p in m.named_parameters():
                if n.startswith('conv_') and n != 'conv_final.weight' and n != 'conv_final.bias':
                    assert p.grad is None, n
                    assert torch.equal(p.detach(), before[n]), n
    for n, g in grads[1].items():
        if g is not None:
            assert torch.allclose(g, grads[0][n], rtol=1e-5, atol=1e-7), n
    assert any(g is not None for g in grads[1].values())
    # a fully frozen model under grad mode is a plain forward (the afterburner script's use)
    for p in m.parameters():
        p.requires_grad = False
    out = m(x)
    assert not out.requires_grad
