#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

Imports ``lib.models.zf_unet``, ``lib.models.tiramisu``, ``lib.losses``, ``lib.metrics`` read-only from
/root/reference (torch 2.10 CPU, fp32) and stores plain arrays (.npz) -- inputs, expected outputs,
never module objects or source.  Weights are torch's default initialisation under a fixed manual_seed (reproduced bit for bit by
oracle.zf_unet_ref.default_init_state), so large configs need no weight payload.

Fixtures
  losses.npz          G6  every binary loss / metric value + d(loss)/d(logits) on a small random tensor,
                          plus hand-derivable known answers
  zf_unet_tiny.npz    G1  ZF_UNET(filters=4, dropout 0) B=2 64x64: eval/train logits, all losses, metrics,
                          every parameter gradient of (B*bce_jaccard).backward(), BN running stats,
                          state after one SGD(1e-3) step, 5-step loss trajectory
  tiramisu_small.npz  G5  small FCDenseNet (reference tiramisu.py) 2x3x36x44: logits, loss, all gradients, buffers
  tiramisu57_small.npz    FCDenseNet57 (growth 12) 2x3x64x64, seeded fill: logits, loss, gradient norms + probes, buffers
  zf_unet_224.npz     G2+G3  ZF_UNET() default (filters=32, Dropout2d 0.2 replay tables captured from the
                          reference's own RNG draw) B=4 224x224: loss / IoU / accuracy scalars, per-tensor
                          gradient L2 norms, probed logits and gradient entries
  unet16_small.npz    the reference's lib/models/unet16.py (UNet16 wiring :52-131) run with a torch.nn stand-in for
                          ``torchvision.models.vgg16`` (cfg "D" features stack): logits, loss, gradient norms + probes
  linknet_small.npz   the reference's lib/models/linknet.py (LinkNet34 wiring :5-90) run with the reference's own
                          ``lib/models/dilated_resnet.py`` (dilated=False) standing in for torchvision's resnet34 and
                          a BatchNorm2d + LeakyReLU stand-in for the un-vendored ``inplace_abn`` backend
  ("wiring pinned, third-party topology restated": what those fixtures pin is the reference's own forward code;
  the torchvision / inplace_abn pieces are restated from their published definitions.)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np
import torch

from lib.models.zf_unet import ZF_UNET          # reference
from lib import losses as ref_losses            # reference
from lib import metrics as ref_metrics          # reference

from oracle import zf_unet_ref, train_step_ref  # only for closed_form_fill / synthetic_batch (input makers)

torch.set_num_threads(8)


def ref_loss(name):
    """torch_train.get_loss (torch_train.py:82-97) + the API-rot fix of SURVEY 8c."""
    if name == 'bce':
        c = ref_losses.BCEWithSigmoidLoss()
        c.size_average, c.reduce = True, True
    elif name == 'bce_jaccard':
        c = ref_losses.BCEWithLogitsLossAndSmoothJaccard()
        c.bce_loss.size_average, c.bce_loss.reduce = True, True
    elif name == 'focal':
        c = ref_losses.FocalLossBinary(size_average=False)
        c.size_average, c.reduce = False, True
    elif name == 'jaccard':
        c = ref_losses.JaccardLoss()
    elif name == 'smooth_jaccard':
        c = ref_losses.SmoothJaccardLoss()
    elif name == 'dice':
        c = ref_losses.DiceLoss()
    elif name == 'bce_dice':
        b, d = ref_loss('bce'), ref_losses.DiceLoss()
        return lambda o, t: (b(o, t) + d(o, t.float())) / 2.0
    else:
        raise ValueError(name)
    if name in ('dice',):
        return lambda o, t: c(o, t.float())
    return c


def _legacy(crit, size_average, reduce):
    """the API-rot fix of SURVEY 8c: torch >= 1.x no longer stores the two legacy attributes the reference reads"""
    crit.size_average, crit.reduce = size_average, reduce
    return crit


LOSS_NAMES = ['bce', 'jaccard', 'smooth_jaccard', 'dice', 'bce_jaccard', 'bce_dice', 'focal']


def gen_losses():
    g = torch.Generator().manual_seed(7)
    x = (2.5 * torch.randn(2, 1, 24, 40, generator=g))
    t = (torch.rand(2, 1, 24, 40, generator=g) > 0.6).long()
    out = {'x': x.numpy(), 't': t.numpy()}
    for name in LOSS_NAMES:
        xx = x.clone().requires_grad_(True)
        l = ref_loss(name)(xx, t)
        (x.shape[0] * l).backward()
        out['loss_' + name] = l.detach().numpy()
        out['dx_' + name] = xx.grad.numpy()
    out['iou'] = ref_metrics.JaccardScore()(x, t).numpy()
    out['acc'] = ref_metrics.PixelAccuracy()(x, t).numpy()
    # the rest of the constructor surface (losses.py:47,84): sum / unreduced BCE, focal with other gammas / mean
    r = torch.randn(x.shape, generator=g)                   # upstream gradient of the unreduced map
    for tag, make, seed_grad in (
            ('bce_sum', lambda: _legacy(ref_losses.BCEWithSigmoidLoss(size_average=False), False, True), None),
            ('bce_none', lambda: _legacy(ref_losses.BCEWithSigmoidLoss(reduce=False), True, False), r),
            ('focal_g15_mean', lambda: _legacy(ref_losses.FocalLossBinary(gamma=1.5), True, True), None),
            ('focal_g3_sum', lambda: _legacy(ref_losses.FocalLossBinary(gamma=3, size_average=False), False, True), None),
            ('focal_g0_sum', lambda: _legacy(ref_losses.FocalLossBinary(gamma=0, size_average=False), False, True), None)):
        xx = x.clone().requires_grad_(True)
        l = make()(xx, t)
        if seed_grad is None:
            (x.shape[0] * l).backward()
        else:
            l.backward(seed_grad)
        out['loss_' + tag] = l.detach().numpy()
        out['dx_' + tag] = xx.grad.numpy()
    out['map_seed'] = r.numpy()
    # known answers: x = 0 everywhere, t = 1 -> bce = log(1.5) + log(2)
    x0 = torch.zeros(1, 1, 4, 4)
    t1 = torch.ones(1, 1, 4, 4).long()
    t0 = torch.zeros(1, 1, 4, 4).long()
    for tag, tt in (('ones', t1), ('zeros', t0)):
        for name in LOSS_NAMES:
            out['ka_%s_%s' % (tag, name)] = ref_loss(name)(x0, tt).numpy()
        out['ka_%s_iou' % tag] = ref_metrics.JaccardScore()(x0, tt).numpy()
    # accuracy edge: nothing matches -> the reference returns an integer zero (metrics.py:37-38)
    xe = torch.full((1, 1, 2, 2), 3.0)
    out['acc_nomatch'] = np.asarray(float(ref_metrics.PixelAccuracy()(xe, t0[:, :, :2, :2])))
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)
    print('losses.npz', {k: float(v) for k, v in out.items() if k.startswith('loss_') and np.ndim(v) == 0})


def make_reference_model(seed, **kw):
    """The reference model with torch's default initialisation under manual_seed(seed).  (An earlier
    closed-form sin() fill made the net pathologically ill-conditioned: the reference's own fp32 loss sat
    1e-5 from its fp64 value; default init sits 4e-8 from it.)  Also checks that the oracle's
    default_init_state reproduces these weights bit for bit."""
    torch.manual_seed(seed)
    m = ZF_UNET(**kw)
    sd = m.state_dict()
    o = zf_unet_ref.default_init_state(filters=kw.get('filters', 32), seed=seed)
    assert list(o.keys()) == list(sd.keys()), 'state_dict layout drifted'
    assert all(torch.equal(o[k], sd[k]) for k in sd), 'default init not reproduced'
    return m


def gen_tiny():
    B, S, F = 2, 64, 4
    m = make_reference_model(3, dropout_val=0.0, filters=F)
    assert list(m.state_dict().keys()) == list(zf_unet_ref.state_shapes(filters=F).keys())
    x, y = train_step_ref.synthetic_batch(B, S, seed=11)
    out = {'x': x.numpy(), 'y': y.numpy()}
    m.eval()
    with torch.no_grad():
        out['eval_logits'] = m(x).numpy()
    m.train()
    logits = m(x)                        # updates running stats once
    out['train_logits'] = logits.detach().numpy()
    for name in LOSS_NAMES:
        out['loss_' + name] = ref_loss(name)(logits.detach(), y).numpy()
    out['iou'] = ref_metrics.JaccardScore()(logits.detach(), y).numpy()
    out['acc'] = ref_metrics.PixelAccuracy()(logits.detach(), y).numpy()
    m.zero_grad()
    l = ref_loss('bce_jaccard')(logits, y)
    (B * l).backward()
    for n, p in m.named_parameters():
        out['grad/' + n] = p.grad.numpy().copy()
    for n, b in m.named_buffers():
        out['buf/' + n] = b.numpy().copy()
    # per-loss gradient norms (full grads only for bce_jaccard above)
    for name in ['bce', 'jaccard', 'dice', 'focal', 'bce_dice']:
        m2 = make_reference_model(3, dropout_val=0.0, filters=F)
        m2.train()
        ll = ref_loss(name)(m2(x), y)
        (B * ll).backward()
        out['gradnorm_' + name] = np.array([p.grad.norm().item() for _, p in m2.named_parameters()])
    # 5-step SGD trajectory from the filled state (fresh model: the forward above moved BN stats)
    m3 = make_reference_model(3, dropout_val=0.0, filters=F)
    m3.train()
    opt = torch.optim.SGD(m3.parameters(), lr=1e-3)
    crit = ref_loss('bce_jaccard')
    traj = []
    for it in range(5):
        opt.zero_grad()
        ll = crit(m3(x), y)
        (B * ll).backward()
        opt.step()
        traj.append(ll.item())
        if it == 0:     # state after ONE step (later steps amplify fp32 noise through the 2x2-pixel BN)
            for n, p in m3.state_dict().items():
                out['after1/' + n] = p.numpy().copy()
    out['traj_bce_jaccard'] = np.array(traj)
    np.savez_compressed(os.path.join(HERE, 'zf_unet_tiny.npz'), **out)
    print('zf_unet_tiny.npz traj', traj)


def capture_dropout_tables(model, p):
    """Forward hooks on every nn.Dropout2d: table[n,c] = 0 where the reference zeroed the channel."""
    tables, hooks = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Dropout2d):
            def hook(_m, inp, outp, name=name):
                i = inp[0].detach().abs().flatten(2).sum(-1)
                o = outp.detach().abs().flatten(2).sum(-1)
                dropped = (o == 0) & (i > 0)
                tables[name.rsplit('.', 1)[0]] = torch.where(dropped, torch.zeros_like(o),
                                                             torch.full_like(o, 1.0 / (1.0 - p)))
            hooks.append(mod.register_forward_hook(hook))
    return tables, hooks


def gen_224():
    B, S = 4, 224
    m = make_reference_model(1)         # defaults: filters=32, dropout 0.2, BN
    x, y = train_step_ref.synthetic_batch(B, S, seed=1234)
    m.train()
    torch.manual_seed(2024)             # the Dropout2d draw
    tables, hooks = capture_dropout_tables(m, 0.2)
    logits = m(x)
    for h in hooks:
        h.remove()
    l = ref_loss('bce_jaccard')(logits, y)
    m.zero_grad()
    (B * l).backward()
    out = {'loss_bce_jaccard': l.detach().numpy(),
           'loss_bce': ref_loss('bce')(logits.detach(), y).numpy(),
           'loss_bce_dice': ref_loss('bce_dice')(logits.detach(), y).numpy(),
           'iou': ref_metrics.JaccardScore()(logits.detach(), y).numpy(),
           'acc': ref_metrics.PixelAccuracy()(logits.detach(), y).numpy()}
    for k, v in tables.items():
        out['drop/' + k] = v.numpy()
    rng = np.random.RandomState(5)
    flat = logits.detach().numpy().reshape(-1)
    idx = rng.choice(flat.size, 2048, replace=False)
    out['logit_idx'] = idx
    out['logit_val'] = flat[idx]
    out['logit_mean'] = np.asarray(flat.mean())
    out['logit_std'] = np.asarray(flat.std())
    pnames, norms = [], []
    for n, p in m.named_parameters():
        g = p.grad.numpy().reshape(-1)
        pnames.append(n)
        norms.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
        k = min(64, g.size)
        gi = rng.choice(g.size, k, replace=False)
        out['gidx/' + n] = gi
        out['gval/' + n] = g[gi]
    out['grad_names'] = np.array(pnames)
    out['grad_norms'] = np.array(norms)
    for n, b in m.named_buffers():
        if 'conv_224' in n or 'conv_7.' in n or 'up_conv_224' in n:
            out['buf/' + n] = b.numpy().copy()
    # the same step by the reference in float64 (Dropout2d modules swapped for the captured tables):
    # tells how far the reference's OWN fp32 arithmetic sits from the exact value
    class _Replay(torch.nn.Module):
        def __init__(self, table):
            super().__init__()
            self.table = table

        def forward(self, t):
            return t * self.table[:, :, None, None]
    m64 = make_reference_model(1).double().train()
    for k, v in tables.items():
        getattr(m64, k).dropout = _Replay(v.double())
    with torch.no_grad():
        l64 = m64(x.double())
    out['loss_bce_jaccard_fp64'] = ref_loss('bce_jaccard')(l64, y).numpy()
    out['iou_fp64'] = ref_metrics.JaccardScore()(l64, y).numpy()
    out['logit_val_fp64'] = l64.numpy().reshape(-1)[idx]
    print('fp32 - fp64 loss of the reference itself: %.3e' % (float(l) - float(out['loss_bce_jaccard_fp64'])))
    np.savez_compressed(os.path.join(HERE, 'zf_unet_224.npz'), **out)
    print('zf_unet_224.npz loss', float(l), 'iou', float(out['iou']))


TIRAMISU_CFG = dict(in_channels=3, down_blocks=(2, 3), up_blocks=(3, 2), bottleneck_layers=2, growth_rate=8,
                    out_chans_first_conv=16, n_classes=1)


def gen_tiramisu():
    """G5: a small FCDenseNet (same block structure as FCDenseNet103, growth 8) on 2x3x36x44: odd pooled sizes
    (36 -> 18 -> 9, 44 -> 22 -> 11) exercise MaxPool floor mode and center_crop of the 2h+1 ConvTranspose
    output.  Dropout2d p set to 0 on the reference instance."""
    from lib.models.tiramisu import FCDenseNet
    torch.manual_seed(11)
    m = FCDenseNet(**TIRAMISU_CFG)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 36, 44, generator=g)
    y = (torch.rand(2, 1, 36, 44, generator=g) > 0.7).long()
    out = {'x': x.numpy(), 'y': y.numpy()}
    for k, v in m.state_dict().items():
        out['sd/' + k] = v.numpy().copy()
    m.train()
    logits = m(x)
    out['train_logits'] = logits.detach().numpy()
    l = ref_loss('bce_jaccard')(logits, y)
    out['loss_bce_jaccard'] = l.detach().numpy()
    out['iou'] = ref_metrics.JaccardScore()(logits.detach(), y).numpy()
    (2 * l).backward()
    for n, p in m.named_parameters():
        out['grad/' + n] = p.grad.numpy().copy()
    for n, b in m.named_buffers():
        out['buf/' + n] = b.numpy().copy()
    m.eval()
    with torch.no_grad():
        out['eval_logits'] = m(x).numpy()
    np.savez_compressed(os.path.join(HERE, 'tiramisu_small.npz'), **out)
    print('tiramisu_small.npz loss', float(l), 'params', sum(p.numel() for p in m.parameters()))


def gen_tiles():
    """lib/tiles.py: pyramid weights, margins / crops and merge() of several image shapes, plus the D4 TTA pair of
    lib/augmentations.py:476-511 (copied as DATA: inputs and outputs).  lib.tiles imports cv2 at module level and
    cv2 is not installed: an EMPTY placeholder module (one constant, no function) makes the import succeed; split()
    and cut_patch(), the only cv2 users, are therefore NOT exercised here."""
    import types
    cv2 = types.ModuleType('cv2')
    cv2.BORDER_REFLECT101 = 4
    sys.modules.setdefault('cv2', cv2)
    from lib import tiles as ref_tiles
    out = {}
    for k, (w, h) in enumerate([(8, 8), (16, 12), (33, 33), (64, 64)]):
        W, Dc, De = ref_tiles.compute_patch_weight_loss(w, h)
        out['pw%d/wh' % k] = np.array([w, h])
        out['pw%d/W' % k], out['pw%d/Dc' % k], out['pw%d/De' % k] = W, Dc, De
    cases = [((60, 75, 3), 32, 16, 0, 'pyramid'), ((64, 64), 32, 32, 0, 'mean'), ((70, 90, 1), 40, 20, 5, 'mean'),
             ((97, 120, 2), 32, 16, 0, 'pyramid')]
    rng = np.random.RandomState(7)
    for k, (shape, ts, step, margin, weight) in enumerate(cases):
        sl = ref_tiles.ImageSlicer(shape, ts, step, margin, weight)
        ch = 1 if len(shape) == 2 else shape[2]
        tiles = rng.rand(len(sl.crops), ts, ts, ch).astype(np.float32)
        out['sl%d/args' % k] = np.array(list(shape) + [0] * (3 - len(shape)) + [len(shape), ts, step, margin,
                                                                                 weight == 'pyramid'])
        out['sl%d/crops' % k] = np.array(sl.crops)
        out['sl%d/margins' % k] = np.array([sl.margin_left, sl.margin_right, sl.margin_top, sl.margin_bottom])
        out['sl%d/tiles' % k] = tiles
        out['sl%d/merged' % k] = sl.merge(list(tiles), dtype=np.float32)      # (2-D tiles break the reference's merge)
    np.savez_compressed(os.path.join(HERE, 'tiles.npz'), **out)
    print('tiles.npz', len(out), 'arrays')


def _install_cv2_standin():
    """cv2 is absent from the image.  What lib.tiles / lib.augmentations need from it at IMPORT time are enum constants
    used as default arguments; what ImageSlicer.split / cut_patch need at run time is copyMakeBorder with
    BORDER_REFLECT101, which is numpy's 'reflect' padding.  Anything else raises."""
    import types

    class _CV2(types.ModuleType):
        BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
        BORDER_REFLECT101 = BORDER_DEFAULT = 4

        def __getattr__(self, name):
            if name.isupper():                 # INTER_*, COLOR_*: only ever stored as default arguments here
                return hash(name) & 0xffff
            raise AttributeError('cv2.%s is not available in this container' % name)

        @staticmethod
        def copyMakeBorder(image, top, bottom, left, right, borderType=4, value=0):
            assert borderType == 4, 'only BORDER_REFLECT101 (the reference default) is restated'
            pad = [(top, bottom), (left, right)] + [(0, 0)] * (image.ndim - 2)
            return np.pad(image, pad, mode='reflect')

    sys.modules['cv2'] = _CV2('cv2')


def gen_augment():
    """What was still a restatement after round 2, now produced by the reference's own code:
    NormalizeImage (lib/augmentations.py:452-460), tta_d4_aug / tta_d4_deaug (:476-511) on NON-symmetric arrays, and
    ImageSlicer.split / cut_patch (lib/tiles.py:98-135) -- the reference's margin / crop / indexing code around a
    copyMakeBorder that is np.pad(mode='reflect') (cv2 itself is absent: _install_cv2_standin)."""
    _install_cv2_standin()
    for mod in ('lib.tiles', 'lib.augmentations'):
        sys.modules.pop(mod, None)
    from lib import augmentations as ref_aug
    from lib import tiles as ref_tiles
    rng = np.random.RandomState(11)
    out = {}
    img = rng.randint(0, 256, size=(9, 7, 3)).astype(np.uint8)
    out['norm/u8'] = img
    out['norm/default'] = ref_aug.NormalizeImage()(img)
    out['norm/custom_args'] = np.array([1. / 128., 0.1, 0.2, 0.3, 0.5, 0.25, 2.0])
    out['norm/custom'] = ref_aug.NormalizeImage(1. / 128., [0.1, 0.2, 0.3], [0.5, 0.25, 2.0])(img)
    out['norm/f32_in'] = ref_aug.NormalizeImage()(img.astype(np.float32))
    sq = [rng.rand(6, 6, 2).astype(np.float32), rng.rand(5, 5).astype(np.float32)]
    for k, a in enumerate(sq):
        out['d4/in%d' % k] = a
        aug = ref_aug.tta_d4_aug([a])
        out['d4/aug%d' % k] = np.stack(aug)
        preds = [rng.rand(*a.shape).astype(np.float32) for _ in range(8)]
        out['d4/preds%d' % k] = np.stack(preds)
        out['d4/deaug%d' % k] = ref_aug.tta_d4_deaug(preds)[0]
    cases = [((37, 45, 3), 16, 8, 0), ((40, 40), 16, 8, 0), ((30, 50, 1), 20, 10, 5), ((33, 33, 2), 32, 16, 0)]
    for k, (shape, ts, step, margin) in enumerate(cases):
        sl = ref_tiles.ImageSlicer(shape, ts, step, margin)
        image = rng.rand(*shape).astype(np.float32)
        tiles = sl.split(image)
        out['split%d/args' % k] = np.array(list(shape) + [0] * (3 - len(shape)) + [len(shape), ts, step, margin])
        out['split%d/image' % k] = image
        out['split%d/tiles' % k] = np.stack(tiles)
        idx = len(tiles) // 2
        out['split%d/patch_index' % k] = np.array(idx)
        out['split%d/patch' % k] = sl.cut_patch(image, idx)
    np.savez_compressed(os.path.join(HERE, 'augment.npz'), **out)
    print('augment.npz', len(out), 'arrays')


# ---------------------------------------------------------------------------------------------------------------------
# UNet16 / LinkNet34: the reference's wiring code run on torch.nn stand-ins for its absent third-party imports
# ---------------------------------------------------------------------------------------------------------------------
def _install_third_party_standins():
    """``torchvision.models.{vgg16, resnet34}`` and ``lib.modules.abn.InPlaceABN`` as torch.nn-only stand-ins.

    * vgg16(...).features: the cfg-"D" stack of torchvision's VGG16 (13 conv3x3 p1 + ReLU, MaxPool2d(2,2) after
      convs 2, 4, 7, 10, 13) -- unet16.py:73-102 addresses it by the indices 0,2,5,7,...,28 this produces.
    * resnet34(...): the reference's OWN restatement of the torchvision ResNet (lib/models/dilated_resnet.py:23-56,
      136-191, imported from /root/reference) with BasicBlock x [3,4,6,3], dilated=False: conv1/bn1/relu/maxpool/
      layer1..4, the attributes linknet.py:41-48 reads.
    * InPlaceABN: BatchNorm2d semantics + LeakyReLU(slope) as bn.py:47-103 / functions.py:62-122 describe them
      (batch mean / biased variance in training, running_var updated with the unbiased variance, standard affine
      gamma -- SURVEY 8c; no num_batches_tracked buffer).
    """
    import types
    import torch.nn as nn
    import torch.nn.functional as F
    from lib.models import dilated_resnet as ref_resnet        # reference

    def vgg16(pretrained=False, **kw):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    def resnet34(pretrained=False, **kw):
        return ref_resnet.DilatedResNet(ref_resnet.BasicBlock, [3, 4, 6, 3], dilated=False)

    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    tv.models.vgg16, tv.models.resnet34 = vgg16, resnet34
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tv.models

    class InPlaceABN(nn.Module):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation='leaky_relu', slope=0.01):
            super().__init__()
            self.eps, self.momentum, self.activation, self.slope = eps, momentum, activation, slope
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
            self.register_buffer('running_mean', torch.zeros(num_features))
            self.register_buffer('running_var', torch.ones(num_features))

        def forward(self, x):
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                             self.momentum, self.eps)
            return F.leaky_relu(y, self.slope) if self.activation == 'leaky_relu' else y

    abn = types.ModuleType('lib.modules.abn')
    abn.InPlaceABN = InPlaceABN
    sys.modules['lib.modules.abn'] = abn


def _run_and_record(m, x, y, fname, seed):
    """Load the seeded fill, run train forward + (B*bce_jaccard).backward() + eval forward, store small arrays."""
    from oracle import fill
    sd = fill.seeded_state(m.state_dict(), seed)
    m.load_state_dict(sd)
    B = x.shape[0]
    out = {'x': x.numpy(), 'y': y.numpy(), 'seed': np.asarray(seed),
           'sd_keys': np.array(list(sd.keys())), 'sd_numel': np.array([v.numel() for v in sd.values()])}
    m.eval()
    with torch.no_grad():
        out['eval_logits'] = m(x).numpy()
    m.train()
    logits = m(x)
    out['train_logits'] = logits.detach().numpy()
    l = ref_loss('bce_jaccard')(logits, y)
    out['loss_bce_jaccard'] = l.detach().numpy()
    out['iou'] = ref_metrics.JaccardScore()(logits.detach(), y).numpy()
    out['acc'] = ref_metrics.PixelAccuracy()(logits.detach(), y).numpy()
    m.zero_grad()
    (B * l).backward()
    rng = np.random.RandomState(9)
    names, norms = [], []
    for n, p in m.named_parameters():
        g = p.grad.numpy().reshape(-1)
        names.append(n)
        norms.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
        gi = rng.choice(g.size, min(64, g.size), replace=False)
        out['gidx/' + n], out['gval/' + n] = gi, g[gi]
    out['grad_names'], out['grad_norms'] = np.array(names), np.array(norms)
    for n, b in m.named_buffers():
        if b.numel() <= 1024:
            out['buf/' + n] = b.numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, 'loss', float(l), 'iou', float(out['iou']), 'params', sum(p.numel() for p in m.parameters()))


def gen_unet16():
    _install_third_party_standins()
    from lib.models.unet16 import UNet16                        # reference wiring, unet16.py:52-131
    m = UNet16(num_classes=1, num_filters=8, pretrained=False)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 64, 96, generator=g)
    y = (torch.rand(2, 1, 64, 96, generator=g) > 0.7).long()
    _run_and_record(m, x, y, 'unet16_small.npz', seed=16)


def gen_unet16_bilinear():
    """The reference's UNet16 (unet16.py:52-131) with every decoder block replaced by the reference's OWN
    DecoderBlock(..., is_deconv=False) (unet16.py:42-46: Upsample(x2, bilinear) -> ConvRelu -> ConvRelu) -- UNet16's constructor
    passes no flag (:104-108), so the blocks are swapped after construction, same channel arguments."""
    _install_third_party_standins()
    import warnings
    from lib.models.unet16 import DecoderBlock, UNet16
    nf = 8
    m = UNet16(num_classes=1, num_filters=nf, pretrained=False)
    m.center = DecoderBlock(512, nf * 8 * 2, nf * 8, is_deconv=False)
    m.dec5 = DecoderBlock(512 + nf * 8, nf * 8 * 2, nf * 8, is_deconv=False)
    m.dec4 = DecoderBlock(512 + nf * 8, nf * 8 * 2, nf * 8, is_deconv=False)
    m.dec3 = DecoderBlock(256 + nf * 8, nf * 4 * 2, nf * 2, is_deconv=False)
    m.dec2 = DecoderBlock(128 + nf * 2, nf * 2 * 2, nf, is_deconv=False)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 3, 64, 96, generator=g)
    y = (torch.rand(2, 1, 64, 96, generator=g) > 0.7).long()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')          # (nn.Upsample's align_corners notice)
        _run_and_record(m, x, y, 'unet16_bilinear_small.npz', seed=17)


def gen_linknet():
    _install_third_party_standins()
    from lib.models.linknet import LinkNet34                    # reference wiring, linknet.py:5-90
    m = LinkNet34(num_classes=1, num_channels=3, pretrained=False)
    m.finaldrop1.p = 0.0
    g = torch.Generator().manual_seed(34)
    x = torch.randn(2, 3, 128, 160, generator=g)
    y = (torch.rand(2, 1, 128, 160, generator=g) > 0.7).long()
    _run_and_record(m, x, y, 'linknet_small.npz', seed=34)


def gen_tiramisu57():
    """FCDenseNet57 (tiramisu.py:187-191: growth rate 12 -- NOT a multiple of 8 -- 4 layers per block) at 2x3x64x64 with the
    seeded fill, Dropout2d off: the reference constructor the product pads slice by slice (12 -> 16 channels)."""
    from lib.models.tiramisu import FCDenseNet57
    m = FCDenseNet57(n_classes=1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    g = torch.Generator().manual_seed(57)
    x = torch.randn(2, 3, 64, 64, generator=g)
    y = (torch.rand(2, 1, 64, 64, generator=g) > 0.7).long()
    _run_and_record(m, x, y, 'tiramisu57_small.npz', seed=57)


if __name__ == '__main__':
    which = sys.argv[1:] or ['losses', 'tiny', '224', 'tiramisu', 'tiramisu57', 'tiles', 'augment', 'unet16', 'unet16_bilinear', 'linknet']
    if 'augment' in which:
        gen_augment()
    if 'unet16' in which:
        gen_unet16()
    if 'unet16_bilinear' in which:
        gen_unet16_bilinear()
    if 'linknet' in which:
        gen_linknet()
    if 'tiles' in which:
        gen_tiles()
    if 'losses' in which:
        gen_losses()
    if 'tiny' in which:
        gen_tiny()
    if '224' in which:
        gen_224()
    if 'tiramisu' in which:
        gen_tiramisu()
    if 'tiramisu57' in which:
        gen_tiramisu57()
