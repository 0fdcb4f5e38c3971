"""Segmentation metrics on the fused HIP reduction -- drop-in for the reference's ``lib.metrics``
(/root/reference/lib/metrics.py:9-43): soft IoU on sigmoid probabilities (eps 1e-7) and pixel accuracy at
threshold 0.5.  Both come out of ONE pass over (logits, target) and stay on the device as 0-dim tensors
(the reference calls ``.cpu().item()`` on them itself, torch_train.py:209-210)."""
from torch.nn.modules.loss import _Loss

from segnb.seglosses import seg_metrics


class JaccardScore(_Loss):
    def __init__(self):
        super(JaccardScore, self).__init__()

    def forward(self, output, target):
        return seg_metrics(output, target)[0]

    def __str__(self):
        return 'JaccardScore'


class PixelAccuracy(_Loss):
    def __init__(self):
        super(PixelAccuracy, self).__init__()

    def forward(self, output, target):
        # the reference returns an integer 0 tensor when nothing matches (metrics.py:37-38); a float 0.0
        # has the same .item() value
        return seg_metrics(output, target)[1]

    def __str__(self):
        return 'PixelAccuracy'
