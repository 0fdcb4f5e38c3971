"""CPU restatement of UNet16 / TernausNet-16 (lib/models/unet16.py:52-131).  TEST INFRASTRUCTURE.

PARITY: WIRING PINNED, THIRD-PARTY TOPOLOGY RESTATED.  tests/golden/unet16_small.npz was produced by running the
reference's own lib/models/unet16.py:52-131 (make_golden.py gen_unet16) with a torch.nn stand-in for the one symbol
it takes from the absent torchvision, ``models.vgg16(...).features`` (the cfg-"D" conv stack, restated from its
published definition); tests/test_models_cpu.py checks this file against it (logits 1e-4, loss 1e-6, gradients
1e-3).  The topology restated here:
VGG16 "D" convs at features indices 0,2 | 5,7 | 10,12,14 | 17,19,21 | 24,26,28, each followed by ReLU
(unet16.py:73-102), MaxPool2d(2,2) between groups (:64,:113-118), DecoderBlock = conv3x3+ReLU ->
ConvTranspose2d(4, stride 2, pad 1) -> ReLU (:35-40), concatenations [decoder, encoder] (:122-127), dec1 =
conv3x3+ReLU, final 1x1 (:111), all from torch.nn.functional on a state_dict with the reference's key names.
"""
import torch
import torch.nn.functional as F

ENC_IDX = [[0, 2], [5, 7], [10, 12, 14], [17, 19, 21], [24, 26, 28]]


def forward(sd, x, is_deconv=True):
    """is_deconv=False: every DecoderBlock on its other branch (unet16.py:42-46): Upsample(scale_factor=2, mode='bilinear') ->
    conv3x3 + ReLU -> conv3x3 + ReLU (keys block.1.conv / block.2.conv); pinned by tests/golden/unet16_bilinear_small.npz, which
    the reference's own DecoderBlock(is_deconv=False) instances produced inside the reference's UNet16."""
    def cr(prefix, h):
        return torch.relu(F.conv2d(h, sd[prefix + 'weight'], sd[prefix + 'bias'], padding=1))

    def dec(name, h):
        if not is_deconv:
            h = F.interpolate(h, scale_factor=2, mode='bilinear', align_corners=False)
            return cr(name + '.block.2.conv.', cr(name + '.block.1.conv.', h))
        h = cr(name + '.block.0.conv.', h)
        return torch.relu(F.conv_transpose2d(h, sd[name + '.block.1.weight'], sd[name + '.block.1.bias'], stride=2,
                                             padding=1))
    skips, h = [], x
    for grp in ENC_IDX:
        for i in grp:
            h = cr('encoder.%d.' % i, h)
        skips.append(h)
        h = F.max_pool2d(h, 2, 2)
    d = dec('center', h)
    for name, k in (('dec5', 4), ('dec4', 3), ('dec3', 2), ('dec2', 1)):
        d = dec(name, torch.cat([d, skips[k]], 1))
    d = cr('dec1.conv.', torch.cat([d, skips[0]], 1))
    return F.conv2d(d, sd['final.weight'], sd['final.bias'])
