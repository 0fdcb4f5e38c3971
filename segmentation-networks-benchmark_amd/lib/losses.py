"""Binary segmentation losses on the fused HIP loss kernels -- drop-in for the reference's ``lib.losses``
(binary part, /root/reference/lib/losses.py:7-101; the multi-class losses at :105-232 are never selected
by ``get_loss``, torch_train.py:82-97, and are out of scope).

Class names, constructor signatures and numerics follow the reference, including its quirks:
  * ``BCEWithSigmoidLoss`` applies the sigmoid twice (logsigmoid, then BCE-with-logits; losses.py:51-53)
  * Jaccard / Dice sums run over the WHOLE batch tensor, not per image (losses.py:13-14,23-24,39-40)
  * ``FocalLossBinary(size_average=False)`` is a SUM (torch_train.py:92)
Each forward is two kernel launches (global sums + finalize), backward one; nothing syncs the host.
"""
from torch.nn.modules.loss import _Loss

from segnb.seglosses import make_spec, seg_loss, seg_loss_map


class DiceLoss(_Loss):
    def __init__(self):
        super(DiceLoss, self).__init__()
        self._spec = make_spec(w_dice=1.0)

    def forward(self, output, target):
        return seg_loss(output, target, self._spec)


class JaccardLoss(_Loss):
    def __init__(self):
        super(JaccardLoss, self).__init__()
        self._spec = make_spec(w_jaccard=1.0)

    def forward(self, output, target):
        return seg_loss(output, target, self._spec)


class SmoothJaccardLoss(_Loss):
    def __init__(self, smooth=100):
        super(SmoothJaccardLoss, self).__init__()
        self.smooth = smooth

    def forward(self, output, target):
        return seg_loss(output, target, make_spec(w_sjaccard=1.0, smooth=self.smooth))


class BCEWithSigmoidLoss(_Loss):
    def __init__(self, size_average=True, reduce=True):
        super(BCEWithSigmoidLoss, self).__init__()
        self.size_average, self.reduce = size_average, reduce

    def forward(self, outputs, targets):
        # the legacy (size_average, reduce) pair of F.binary_cross_entropy_with_logits (losses.py:53):
        # reduce=False -> per-pixel map; else mean (size_average) or sum
        if not self.reduce:
            return seg_loss_map(outputs, targets, 0)
        return seg_loss(outputs, targets, make_spec(w_bce=1.0, bce_sum=0 if self.size_average else 1))


class BCEWithLogitsLossAndSmoothJaccard(_Loss):
    """(bce_weight * BCE + jaccard_weight * SmoothJaccard) / (bce_weight + jaccard_weight), arXiv:1706.06169"""

    def __init__(self, bce_weight=1, jaccard_weight=0.5):
        super(BCEWithLogitsLossAndSmoothJaccard, self).__init__()
        self.bce_loss = BCEWithSigmoidLoss()
        self.jac_loss = SmoothJaccardLoss()
        self.bce_weight = bce_weight
        self.jaccard_weight = jaccard_weight

    def forward(self, outputs, targets):
        spec = make_spec(w_bce=self.bce_weight, w_sjaccard=self.jaccard_weight, smooth=self.jac_loss.smooth,
                         norm=self.bce_weight + self.jaccard_weight)
        return seg_loss(outputs, targets, spec)


class BCEAndDiceLoss(_Loss):
    """(bce_weight * BCE + dice_weight * Dice) / (bce_weight + dice_weight).  Build-defined: BASELINE.json
    config 2 names "BCE+Dice"; the reference ships both terms (losses.py:7,46) but no wired combination."""

    def __init__(self, bce_weight=1, dice_weight=1):
        super(BCEAndDiceLoss, self).__init__()
        self.bce_weight, self.dice_weight = bce_weight, dice_weight

    def forward(self, outputs, targets):
        spec = make_spec(w_bce=self.bce_weight, w_dice=self.dice_weight, norm=self.bce_weight + self.dice_weight)
        return seg_loss(outputs, targets, spec)


class FocalLossBinary(_Loss):
    def __init__(self, gamma=2, size_average=True, reduce=True):
        super(FocalLossBinary, self).__init__()
        self.gamma = gamma
        self.size_average, self.reduce = size_average, reduce

    def forward(self, outputs, targets):
        # `reduce` is accepted and ignored, as in the reference (losses.py:97-101 always reduces)
        return seg_loss(outputs, targets, make_spec(w_focal=1.0, focal_mean=1 if self.size_average else 0,
                                                    focal_gamma=self.gamma))
