"""-m gpu: teacher-forced plan parity (tests/abi_replay.py) -- every C-ABI call of one real training step of each
model family, HIP kernel vs ABI emulator on the emulator's own inputs, at per-op tolerance:
    f32 : 2e-4 of the tensor maximum (exact-fp32 MFMA, summation order only)
    bf16: 2e-2 (identical bf16 storage rounding on both sides; 1-2 bf16 ulps after re-rounding)
BatchNorm/activation calls may differ in at most 3 elements per tensor (a ReLU / max-pool argmax whose argument is
within an ulp of the decision point)."""
import os
import warnings

import numpy as np
import pytest
import torch

import abi_replay
import model_checks as mc
from oracle import train_step_ref, zf_unet_ref

pytestmark = pytest.mark.gpu


def _loss():
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    return BCEWithLogitsLossAndSmoothJaccard()


def _report(n, rep):
    assert n > 50
    assert not rep, '%d of %d calls differ:\n%s' % (len(rep), n, '\n'.join(rep[:20]))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_replay_zf_unet_odd_filters_with_dropout(dtype):
    from lib.models.zf_unet import ZF_UNET
    B, S, F = 2, 64, 6
    x, y = train_step_ref.synthetic_batch(B, S, seed=5)
    drop = zf_unet_ref.make_dropout_tables(F, B, 0.2, torch.Generator().manual_seed(3))

    def make():
        torch.manual_seed(2)
        m = ZF_UNET(dropout_val=0.2, filters=F)
        m.dropout_override = drop
        return m
    from lib.losses import BCEAndDiceLoss
    _report(*abi_replay.replay(make, x, y, BCEAndDiceLoss(), dtype))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_replay_zf_unet_default_width(dtype):
    from lib.models.zf_unet import ZF_UNET
    x, y = train_step_ref.synthetic_batch(2, 96, seed=6)

    def make():
        torch.manual_seed(4)
        return ZF_UNET(dropout_val=0.0)
    _report(*abi_replay.replay(make, x, y, _loss(), dtype))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_replay_unet16(dtype):
    _, _, x, y = mc.make_unet16()
    _report(*abi_replay.replay(lambda: mc.make_unet16()[0], x, y, _loss(), dtype))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_replay_linknet34(dtype):
    _, _, x, y = mc.make_linknet(64)
    _report(*abi_replay.replay(lambda: mc.make_linknet(64)[0], x, y, _loss(), dtype))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_replay_tiramisu(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, 'tiramisu_small.npz'))
    _, _, x, y = mc.make_tiramisu(g)
    _report(*abi_replay.replay(lambda: mc.make_tiramisu(g)[0], x, y, _loss(), dtype))
