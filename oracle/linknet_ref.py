"""CPU restatement of LinkNet34 (lib/models/linknet.py:5-90).  TEST INFRASTRUCTURE.

PARITY: WIRING PINNED, THIRD-PARTY TOPOLOGY RESTATED.  tests/golden/linknet_small.npz was produced by running the
reference's own lib/models/linknet.py:5-90 (make_golden.py gen_linknet) with the reference's importable
lib/models/dilated_resnet.py (BasicBlock x [3,4,6,3], dilated=False) standing in for torchvision's resnet34 and a
BatchNorm2d + LeakyReLU(0.01) module standing in for the un-vendored ``inplace_abn`` backend (bn.py:47-103,
functions.py:62-122; gamma-vs-|gamma| of that backend stays undeterminable from the reference);
tests/test_models_cpu.py checks this file against it (logits 1e-4, loss 1e-6, gradients 1e-3).  Restated here:
  stem  conv7x7 s2 p3 (no bias) -> BN -> ReLU -> MaxPool 3x3 s2 p1                          linknet.py:41-44,67-70
  encoder  torchvision BasicBlock x [3,4,6,3] (conv3x3-BN-ReLU-conv3x3-BN (+1x1 s2 downsample-BN) add ReLU) :45-48
  decoder  conv1x1 -> ABN -> ConvTranspose 4x4 s2 p1 -> ABN -> conv1x1 -> ABN, ABN = BN + LeakyReLU(0.01)   :12-30
  skips are ADDs :77-79;  head Dropout2d(.5) -> ConvT 3x3 s2 -> LeakyReLU -> conv3x3 p0 -> LeakyReLU -> conv2x2 p1  :57-62
InPlaceABN affine taken as the standard gamma.
"""
import torch
import torch.nn.functional as F

LAYERS = [3, 4, 6, 3]


def _bn(sd, p, x, train):
    y = F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                     training=train, momentum=0.1, eps=1e-5)
    if train and (p + 'num_batches_tracked') in sd:
        sd[p + 'num_batches_tracked'] += 1
    return y


def forward(sd, x, train=True, drop=None):
    h = torch.relu(_bn(sd, 'firstbn.', F.conv2d(x, sd['firstconv.weight'], None, stride=2, padding=3), train))
    h = F.max_pool2d(h, 3, 2, 1)
    feats = []
    for li, n in enumerate(LAYERS):
        for bi in range(n):
            p = 'encoder%d.%d.' % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            a = torch.relu(_bn(sd, p + 'bn1.', F.conv2d(h, sd[p + 'conv1.weight'], None, stride=stride, padding=1), train))
            b = _bn(sd, p + 'bn2.', F.conv2d(a, sd[p + 'conv2.weight'], None, padding=1), train)
            ident = h
            if (p + 'downsample.0.weight') in sd:
                ident = _bn(sd, p + 'downsample.1.', F.conv2d(h, sd[p + 'downsample.0.weight'], None, stride=stride), train)
            h = torch.relu(b + ident)
        feats.append(h)
    e1, e2, e3, e4 = feats

    def abn(p, t):
        return F.leaky_relu(_bn(sd, p, t, train), 0.01)

    def dec(name, t):
        t = abn(name + '.abn1.', F.conv2d(t, sd[name + '.conv1.weight'], sd[name + '.conv1.bias']))
        t = abn(name + '.abn2.', F.conv_transpose2d(t, sd[name + '.deconv2.weight'], sd[name + '.deconv2.bias'],
                                                    stride=2, padding=1))
        return abn(name + '.abn3.', F.conv2d(t, sd[name + '.conv3.weight'], sd[name + '.conv3.bias']))
    d4 = dec('decoder4', e4) + e3
    d3 = dec('decoder3', d4) + e2
    d2 = dec('decoder2', d3) + e1
    d1 = dec('decoder1', d2)
    if drop is not None:
        d1 = d1 * drop[:, :, None, None]
    f = F.leaky_relu(F.conv_transpose2d(d1, sd['finaldeconv1.weight'], sd['finaldeconv1.bias'], stride=2), 0.01)
    f = F.leaky_relu(F.conv2d(f, sd['finalconv2.weight'], sd['finalconv2.bias']), 0.01)
    return F.conv2d(f, sd['finalconv3.weight'], sd['finalconv3.bias'], padding=1)
