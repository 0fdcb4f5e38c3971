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
