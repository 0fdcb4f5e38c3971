"""CPU restatement of the reference's binary losses and metrics.  TEST INFRASTRUCTURE.

Closed forms, written from the math (not from the reference's module code), in
torch-CPU fp32 so autograd can produce the reference gradients:

* ``bce``            lib/losses.py:46-53   sigmoid applied twice (logsigmoid, then BCE-with-logits)
* ``jaccard``        lib/losses.py:18-28   1 - I/(U - I + 1e-7), sums over the WHOLE batch tensor
* ``smooth_jaccard`` lib/losses.py:31-43   1 - (I+100)/(U - I + 100)
* ``dice``           lib/losses.py:7-15    1 - 2I/(U + 1e-7)
* ``bce_jaccard``    lib/losses.py:56-75   (1*bce + 0.5*smooth_jaccard)/1.5
* ``focal``          lib/losses.py:78-101  sum or mean of (1-pt)^2 * bce_elem, pt = exp(-bce_elem)
* ``jaccard_score``  lib/metrics.py:9-20   I/(U - I + 1e-7)
* ``pixel_accuracy`` lib/metrics.py:26-40  mean((sigmoid(x) > 0.5) == t)

``I = sum(p*t)``, ``U = sum(p) + sum(t)``, ``p = sigmoid(x)``.
Checked against the imported reference by tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F


def _f(t):
    return t.to(torch.float32)


def bce_double_sigmoid_elem(x, t):
    """Per-element value of the reference 'bce': BCEWithLogits(logsigmoid(x), t).

    With z = logsigmoid(x) <= 0:  max(z,0) - z*t + log(1+exp(-|z|)) = -t*z + log(1 + sigmoid(x)).
    """
    z = F.logsigmoid(x)
    return -_f(t) * z + torch.log1p(torch.exp(z))


def bce(x, t):
    return bce_double_sigmoid_elem(x, t).mean()


def _sums(x, t):
    p = torch.sigmoid(x)
    t = _f(t)
    return (p * t).sum(), p.sum() + t.sum()


def jaccard(x, t):
    i, u = _sums(x, t)
    return 1 - i / (u - i + 1e-7)


def smooth_jaccard(x, t, smooth=100.0):
    i, u = _sums(x, t)
    return 1 - (i + smooth) / (u - i + smooth)


def dice(x, t):
    i, u = _sums(x, t)
    return 1 - 2 * i / (u + 1e-7)


def bce_jaccard(x, t, bce_weight=1.0, jaccard_weight=0.5):
    return (bce(x, t) * bce_weight + smooth_jaccard(x, t) * jaccard_weight) / (bce_weight + jaccard_weight)


def bce_dice(x, t, bce_weight=1.0, dice_weight=1.0):
    """Build-defined key (BASELINE.json config 2 'BCE+Dice'); the reference has DiceLoss
    (lib/losses.py:7) and the bce term but no wired combination (torch_train.py:82-97)."""
    return (bce(x, t) * bce_weight + dice(x, t) * dice_weight) / (bce_weight + dice_weight)


def focal(x, t, gamma=2.0, size_average=False):
    e = bce_double_sigmoid_elem(x, t)          # = -logpt
    pt = torch.exp(-e)
    l = (1 - pt).pow(gamma) * e
    return l.mean() if size_average else l.sum()


def jaccard_score(x, t):
    i, u = _sums(x, t)
    return i / (u - i + 1e-7)


def pixel_accuracy(x, t):
    pred = torch.sigmoid(x) > 0.5
    return (pred == (t != 0)).to(torch.float32).mean()


LOSSES = {
    'bce': bce,
    'jaccard': jaccard,
    'smooth_jaccard': smooth_jaccard,
    'dice': dice,
    'bce_jaccard': bce_jaccard,
    'bce_dice': bce_dice,
    'focal': focal,
}
