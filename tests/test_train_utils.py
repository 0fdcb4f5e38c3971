"""lib.train_utils / torch_train harness: host logic vs direct restatements of the reference's arithmetic."""
import numpy as np
import pytest
import torch

from oracle import abi_emulator
from segnb import _native as nv


@pytest.fixture(autouse=True)
def emulated_abi():
    nv.set_backend_for_testing(abi_emulator.AbiEmulator())
    yield
    nv.set_backend_for_testing(None)


def test_average_meter():
    from lib.train_utils import AverageMeter
    m = AverageMeter()
    m.update(2.0)
    m.update(4.0, n=3)
    assert m.val == 4.0 and m.count == 4 and abs(m.avg - 3.5) < 1e-12 and str(m) == '3.500'


def test_pr_curve_meter_matches_threshold_loop():
    """The reference loops over 127 thresholds and bincounts (train_utils.py:109-125); same counts."""
    from lib.train_utils import PRCurveMeter
    g = torch.Generator().manual_seed(0)
    logits = 3 * torch.randn(2, 1, 24, 17, generator=g)
    y = (torch.rand(2, 1, 24, 17, generator=g) > 0.6).long()
    m = PRCurveMeter()
    m.update(logits, y)
    m.update(logits * 0.5, y)
    tp = np.zeros(127, np.uint64)
    tn, fp, fn = tp.copy(), tp.copy(), tp.copy()
    for lg in (logits, logits * 0.5):
        p = torch.sigmoid(lg).numpy().reshape(-1)
        t = y.numpy().reshape(-1).astype(np.int64)
        for i, v in enumerate(np.arange(0., 1., 1. / 127, dtype=np.float32)):
            pred = (p > v).astype(np.int64)
            conf = np.bincount(pred + 2 * t, minlength=4).reshape(2, 2).astype(np.uint64)
            tp[i] += conf[1, 1]
            tn[i] += conf[0, 0]
            fp[i] += conf[0, 1]
            fn[i] += conf[1, 0]
    for a, b in ((m.tp, tp), (m.tn, tn), (m.fp, fp), (m.fn, fn)):
        assert np.array_equal(a, b)
    assert m.precision().shape == (127,) and m.recall().shape == (127,)


def test_pr_curve_meter_accepts_host_tensors_without_the_library():
    """the reference calls .cpu() on both arguments (train_utils.py:111-112), so host tensors are legal input;
    without a device the histogram is taken with torch ops instead of handing host pointers to the kernel."""
    from lib.train_utils import PRCurveMeter
    g = torch.Generator().manual_seed(1)
    logits = 2 * torch.randn(3, 1, 9, 11, generator=g)
    y = (torch.rand(3, 1, 9, 11, generator=g) > 0.5).long()
    emu = PRCurveMeter()
    emu.update(logits, y)                       # through the ABI (emulated here)
    nv.set_backend_for_testing(None)
    host = PRCurveMeter()
    host.update(logits, y.to(torch.uint8))      # host path, other target dtype
    for a, b in ((host.tp, emu.tp), (host.tn, emu.tn), (host.fp, emu.fp), (host.fn, emu.fn)):
        assert np.array_equal(a, b)


def test_recording_is_closed_when_a_recorded_step_raises():
    """an exception inside a step that is being recorded must not leave the thread in recording mode."""
    from lib.models.zf_unet import ZF_UNET, _ZFUnetPlan
    be = nv._test_backend
    state = {'rec': 0}
    be.segnb_plan_begin = lambda: state.__setitem__('rec', state['rec'] + 1) or 0
    def end(h, n):
        state['rec'] -= 1
        return 0
    be.segnb_plan_end = end
    torch.manual_seed(0)
    m = ZF_UNET(dropout_val=0.0, filters=4).set_compute_dtype('f32').train()
    x = torch.randn(1, 3, 32, 32)
    with torch.no_grad():
        m(x)
    eng = m._engine
    eng._cplan_key = lambda *a, **k: ('forced',) + tuple(str(v) for v in a[:4])     # force the recording path on CPU
    orig = be.segnb_head_fwd
    be.segnb_head_fwd = lambda *a: (_ for _ in ()).throw(RuntimeError('boom'))
    with pytest.raises(RuntimeError, match='boom'):
        with torch.no_grad():
            m(x)
    assert state['rec'] == 0 and eng._rec is None
    be.segnb_head_fwd = orig
    with torch.no_grad():
        out = m(x)                               # the configuration is remembered as eager and still runs
    assert out.shape == (1, 1, 32, 32) and state['rec'] == 0


def test_auto_file(tmp_path):
    from lib.train_utils import auto_file
    (tmp_path / 'a' / 'b').mkdir(parents=True)
    (tmp_path / 'a' / 'b' / 'w.pth').write_text('x')
    assert auto_file('w.pth', str(tmp_path)).endswith('a/b/w.pth')
    with pytest.raises(FileNotFoundError):
        auto_file('missing.pth', str(tmp_path))
    (tmp_path / 'a' / 'w.pth').write_text('y')
    with pytest.raises(FileNotFoundError):
        auto_file('w.pth', str(tmp_path))


def test_factories_and_train_validate_loop():
    import torch_train as TT
    for key in ('jaccard', 'bce_jaccard', 'focal', 'bce'):          # the reference's keys (torch_train.py:82-97)
        TT.get_loss(key)
    with pytest.raises(ValueError):
        TT.get_loss('nope')
    with pytest.raises(ValueError):
        TT.get_model('nope')
    with pytest.raises(ValueError):
        TT.get_optimizer('nope', [], 0.1)
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(0)
    model = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    data = [(torch.randn(2, 3, 32, 32), (torch.rand(2, 1, 32, 32) > 0.7).long()) for _ in range(3)]
    opt = TT.get_optimizer('sgd', model.parameters(), 1e-3)
    losses, scores = TT.train(model, TT.get_loss('bce_jaccard'), opt, data, metrics=TT.default_metrics())
    assert losses.count == 3 and np.isfinite(losses.avg) and 0 <= scores['iou'].avg <= 1
    vl, vs = TT.validate(model, TT.get_loss('bce_jaccard'), data, metrics=TT.default_metrics())
    assert vl.count == 3 and np.isfinite(vl.avg)


def test_find_optimal_lr_accumulates_like_reference():
    from lib.models.zf_unet import ZF_UNET
    from lib.losses import BCEWithSigmoidLoss
    from lib.train_utils import find_optimal_lr
    torch.manual_seed(0)
    model = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    batch = (torch.randn(2, 3, 32, 32), (torch.rand(2, 1, 32, 32) > 0.7).long())
    opt = torch.optim.SGD(model.parameters(), lr=1.0)
    lrs, loss = find_optimal_lr(model, BCEWithSigmoidLoss(), opt, [batch] * 30)
    assert lrs.shape == (30,) and abs(lrs[0] - 1e-8) < 1e-12 and abs(lrs[-1] / lrs[0] - 2.0 ** 29) < 1e3
    assert np.all(np.isfinite(loss)) and loss.shape == (30,)


@pytest.mark.parametrize('name', ['sgd', 'rms', 'adam'])
def test_flat_optimizers_match_torch(name):
    """segnb.optim.{SGD,RMSprop,Adam}: one launch over the flat buffers == torch.optim on the same gradients."""
    import copy
    import torch_train as TT
    from lib.models.zf_unet import ZF_UNET
    from lib.losses import BCEWithSigmoidLoss
    torch.manual_seed(0)
    model = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    x, y = torch.randn(2, 3, 32, 32), (torch.rand(2, 1, 32, 32) > 0.7).long()
    opt = TT.get_optimizer(name, model.parameters(), 1e-2)
    ref_params = None
    ref_opt = None
    for it in range(3):
        opt.zero_grad()
        loss = BCEWithSigmoidLoss()(model(x), y)
        (2 * loss).backward()
        if ref_params is None:                  # torch optimizer over detached copies, fed the SAME gradients
            ref_params = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
            ref_opt = {'sgd': torch.optim.SGD, 'rms': torch.optim.RMSprop, 'adam': torch.optim.Adam}[name](ref_params, lr=1e-2)
        for rp, p in zip(ref_params, model.parameters()):
            rp.grad = p.grad.detach().clone()
        opt.step()
        ref_opt.step()
        for rp, p in zip(ref_params, model.parameters()):
            assert torch.allclose(p.detach(), rp.detach(), rtol=1e-5, atol=1e-6), (name, it)
    assert getattr(opt, '_segnb_state', None) is not None or name == 'sgd'


@pytest.mark.parametrize('name', ['rms', 'adam'])
def test_flat_optimizer_state_dict_round_trip(name):
    """optimizer.state_dict() of the one-launch optimizers carries the moments and the step count (views of the flat
    buffers), equals torch.optim's after the same steps, and a run resumed from it continues bit-identically
    (restore_snapshot, torch_train.py:319-330).  ADVICE r1: the state used to live outside optimizer.state."""
    import torch_train as TT
    from lib.models.zf_unet import ZF_UNET
    from lib.losses import BCEWithSigmoidLoss
    x, y = torch.randn(2, 3, 32, 32), (torch.rand(2, 1, 32, 32) > 0.7).long()

    def run(model, opt, steps):
        for _ in range(steps):
            opt.zero_grad()
            (2 * BCEWithSigmoidLoss()(model(x), y)).backward()
            opt.step()

    torch.manual_seed(0)
    a = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    oa = TT.get_optimizer(name, a.parameters(), 1e-2)
    run(a, oa, 2)
    sd = oa.state_dict()
    nparams = len(list(a.parameters()))
    assert len(sd['state']) == nparams
    keys = {'rms': {'step', 'square_avg'}, 'adam': {'step', 'exp_avg', 'exp_avg_sq'}}[name]
    assert set(sd['state'][0].keys()) == keys and float(sd['state'][0]['step']) == 2.0
    # resume in a fresh model + optimizer from (model.state_dict, optimizer.state_dict)
    import copy
    msd, osd = copy.deepcopy(a.state_dict()), copy.deepcopy(sd)
    b = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    b.load_state_dict(msd)
    ob = TT.get_optimizer(name, b.parameters(), 1e-2)
    ob.load_state_dict(osd)
    run(a, oa, 2)
    run(b, ob, 2)
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(pa, pb), n
    assert float(ob.state_dict()['state'][0]['step']) == 4.0
    mom = 'square_avg' if name == 'rms' else 'exp_avg_sq'
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(oa.state[pa][mom], ob.state[pb][mom])


@pytest.mark.parametrize('tag', __import__('model_checks').EXTRA_LOSS_CASES)
def test_loss_constructor_surface_vs_reference_golden(golden_dir, tag):
    """BCEWithSigmoidLoss(size_average=False | reduce=False), FocalLossBinary(gamma != 2): host logic + emulator vs the
    reference's values (VERDICT r1 missing #6)."""
    import os
    import model_checks as mc
    mc.check_extra_loss_case(np.load(os.path.join(golden_dir, 'losses.npz')), tag, 'cpu')


class _CountingBackend(abi_emulator.AbiEmulator):
    def __init__(self):
        self.calls = {}

    def __getattribute__(self, name):
        attr = object.__getattribute__(self, name)
        if name.startswith('segnb_') and callable(attr):
            calls = object.__getattribute__(self, 'calls')
            calls[name] = calls.get(name, 0) + 1
        return attr


def test_metrics_reuse_the_loss_launch():
    """torch_train.py:185,209-210 calls loss, then JaccardScore and PixelAccuracy on the same (outputs, y): ONE
    reduction pass serves all three; different tensors, or a tensor changed in place, reduce again."""
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore, PixelAccuracy
    from oracle import losses_ref
    be = _CountingBackend()
    nv.set_backend_for_testing(be)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, 16, 16, generator=g).requires_grad_(True)
    y = (torch.rand(2, 1, 16, 16, generator=g) > 0.6).long()
    loss = BCEWithLogitsLossAndSmoothJaccard()(x, y)
    iou, acc = JaccardScore()(x, y), PixelAccuracy()(x, y)
    assert be.calls.get('segnb_seg_loss_reduce') == 1
    assert abs(iou.item() - losses_ref.jaccard_score(x.detach(), y).item()) < 1e-6
    assert abs(acc.item() - losses_ref.pixel_accuracy(x.detach(), y).item()) < 1e-7
    loss.backward()
    x2 = x.detach().clone()
    JaccardScore()(x2, y)
    assert be.calls['segnb_seg_loss_reduce'] == 2          # another tensor: its own pass
    PixelAccuracy()(x2, y)
    assert be.calls['segnb_seg_loss_reduce'] == 2          # ... shared by the second metric
    x2.mul_(2.0)
    i2 = JaccardScore()(x2, y)
    assert be.calls['segnb_seg_loss_reduce'] == 3          # changed in place: recomputed
    assert abs(i2.item() - losses_ref.jaccard_score(x2, y).item()) < 1e-6


def test_grad_abs_max_and_snapshot_round_trip(tmp_path):
    """torch_train.py:199-205 (one fused reduction instead of a sync per tensor) and :308-330 (checkpoint dict)."""
    import pandas as pd
    import torch_train as TT
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(0)
    model = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    opt = TT.get_optimizer('adam', model.parameters(), 1e-3)
    data = [(torch.randn(2, 3, 32, 32), (torch.rand(2, 1, 32, 32) > 0.7).long()) for _ in range(2)]
    seen = []
    TT.train(model, TT.get_loss('bce_jaccard'), opt, data, metrics=TT.default_metrics(),
             grad_monitor=lambda step, v: seen.append(v))
    loop = max(p.grad.abs().max().item() for p in model.parameters())
    assert len(seen) == 2 and seen[-1] == loop and TT.grad_global_abs_max(model) == loop
    hist = pd.DataFrame({'epoch': [0], 'train_loss': [0.5]})
    f = str(tmp_path / 'snap.pth')
    TT.save_snapshot(model, opt, 0.5, 3, hist, f)
    ck = torch.load(f, weights_only=False)
    assert sorted(ck.keys()) == ['args', 'epoch', 'loss', 'model', 'optimizer', 'train_history']
    assert len(ck['optimizer']['state']) == len(list(model.parameters()))
    m2 = ZF_UNET(filters=4, dropout_val=0.0).set_compute_dtype('f32')
    o2 = TT.get_optimizer('adam', m2.parameters(), 1e-3)
    start, h2, best = TT.restore_snapshot(m2, o2, f)
    assert start == 4 and best == 0.5 and list(h2.columns) == ['epoch', 'train_loss']
    for (k, a), b in zip(model.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    # the resumed run continues exactly like the uninterrupted one
    TT.train(model, TT.get_loss('bce_jaccard'), opt, data)
    TT.train(m2, TT.get_loss('bce_jaccard'), o2, data)
    for (k, a), b in zip(model.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
