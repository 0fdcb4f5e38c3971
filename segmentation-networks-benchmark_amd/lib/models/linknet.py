"""LinkNet34 on the MI355X engine -- drop-in for the reference's ``lib.models.linknet``
(/root/reference/lib/models/linknet.py:5-90): ``LinkNet34(num_classes=1, num_channels=3, pretrained=True)``, same
attribute tree / state_dict keys (``firstconv.weight``, ``encoder2.0.downsample.0.weight``,
``decoder4.abn1.running_mean``, ``finalconv3.bias`` ...), fp32 NCHW logits.

The reference takes the encoder from ``torchvision.models.resnet34`` (linknet.py:39-48); torchvision is not a
dependency here, so the ResNet34 stack (7x7 s2 stem, BN, ReLU, MaxPool 3x3 s2 p1, BasicBlock x [3,4,6,3]) is built
locally with torchvision's initialisation.  ``pretrained=True`` would download ImageNet weights: not possible
offline -- the model warns and keeps the random initialisation; load a state_dict to use pretrained weights.

Plan (segnb.net): every conv + BatchNorm/InPlaceABN + (residual add) + ReLU/LeakyReLU is one conv_unit; the
stride-2 and 7x7 convs and the ConvTranspose2d layers run on the generalised gather kernel; skip connections are
ADDs (linknet.py:77-79); the 3-layer classifier head ends in a 2x2 conv whose bf16/fp32 NHWC output is converted
to fp32 NCHW logits.
"""
import warnings

import torch
from torch import nn

from lib.modules.abn import InPlaceABN
from segnb import _native as nv
from segnb import convplan as cp
from segnb.net import HipNet, add, conv_unit, head_conv, head_from_act, maxpool


def _holder_forward(self, *a, **k):
    raise RuntimeError('parameter holder; run the whole LinkNet34 (HIP executor)')


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock parameter layout: conv1, bn1, relu, conv2, bn2, downsample."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
    forward = _holder_forward


def _resnet_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes))
    layers = [BasicBlock(inplanes, planes, stride, downsample)]
    layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class DecoderBlockLinkNet(nn.Module):
    def __init__(self, in_channels, n_filters):
        super(DecoderBlockLinkNet, self).__init__()
        self.conv1 = nn.Conv2d(in_channels, in_channels // 4, 1)
        self.abn1 = InPlaceABN(in_channels // 4)
        self.deconv2 = nn.ConvTranspose2d(in_channels // 4, in_channels // 4, kernel_size=4, stride=2, padding=1,
                                          output_padding=0)
        self.abn2 = InPlaceABN(in_channels // 4)
        self.conv3 = nn.Conv2d(in_channels // 4, n_filters, 1)
        self.abn3 = InPlaceABN(n_filters)
    forward = _holder_forward


class LinkNet34(HipNet):
    # the outputs of the encoder's BasicBlocks have two consumers (the next block's first convolution and its identity branch,
    # linknet.py:41-62 via resnet34): their two gradient contributions go to the producing layer's reduction pass as two sources
    # (segnb.net.Tape.lazy_add) instead of through an add pass -- 25 launches and 0.2 ms per step at 512 x 512, bs 16
    lazy_add = True
    def __init__(self, num_classes=1, num_channels=3, pretrained=True):
        super(LinkNet34, self).__init__()
        assert num_channels == 3
        if pretrained:
            warnings.warn('LinkNet34(pretrained=True): ImageNet weights cannot be downloaded here; '
                          'keeping the random initialisation (load_state_dict accepts reference checkpoints)')
        self.num_classes = num_classes
        filters = [64, 128, 256, 512]
        self.firstconv = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.firstbn = nn.BatchNorm2d(64)
        self.firstrelu = nn.ReLU(inplace=True)
        self.firstmaxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.encoder1 = _resnet_layer(64, 64, 3, 1)
        self.encoder2 = _resnet_layer(64, 128, 4, 2)
        self.encoder3 = _resnet_layer(128, 256, 6, 2)
        self.encoder4 = _resnet_layer(256, 512, 3, 2)
        for m in self.modules():                         # torchvision resnet initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        self.decoder4 = DecoderBlockLinkNet(filters[3], filters[2])
        self.decoder3 = DecoderBlockLinkNet(filters[2], filters[1])
        self.decoder2 = DecoderBlockLinkNet(filters[1], filters[0])
        self.decoder1 = DecoderBlockLinkNet(filters[0], filters[0])
        self.finaldrop1 = nn.Dropout2d(p=0.5)
        self.finaldeconv1 = nn.ConvTranspose2d(filters[0], 32, 3, stride=2)
        self.finalrelu1 = nn.LeakyReLU(inplace=True)
        self.finalconv2 = nn.Conv2d(32, 32, 3)
        self.finalrelu2 = nn.LeakyReLU(inplace=True)
        self.finalconv3 = nn.Conv2d(32, num_classes, 2, padding=1)
        self._init_engine(3)

    def _check_input(self, x):
        super(LinkNet34, self)._check_input(x)
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError('LinkNet34 needs H and W divisible by 32, got %dx%d' % (x.shape[2], x.shape[3]))

    def _build(self, tape, x, dlogits):
        def seg(c):
            return [(c, cp.pad8(c))]

        h = conv_unit(tape, x, self.firstconv.weight, None, seg(3), stride=2, pad=3, bn=self.firstbn,
                      act=nv.ACT_RELU, tag='stem')
        h = maxpool(tape, h, 3, 2, 1, tag='stempool')
        feats, c = [], 64
        for li, layer in enumerate((self.encoder1, self.encoder2, self.encoder3, self.encoder4)):
            for bi, blk in enumerate(layer):
                tag = 'enc%d.%d' % (li + 1, bi)
                planes = blk.conv1.out_channels
                a = conv_unit(tape, h, blk.conv1.weight, None, seg(c), stride=blk.stride, pad=1, bn=blk.bn1,
                              act=nv.ACT_RELU, tag=tag + '.c1')
                ident = h
                if blk.downsample is not None:
                    ident = conv_unit(tape, h, blk.downsample[0].weight, None, seg(c), stride=blk.stride, pad=0,
                                      bn=blk.downsample[1], act=nv.ACT_NONE, tag=tag + '.ds')
                h = conv_unit(tape, a, blk.conv2.weight, None, seg(planes), stride=1, pad=1, bn=blk.bn2,
                              act=nv.ACT_RELU, res=ident, tag=tag + '.c2')
                c = planes
            feats.append(h)
        e1, e2, e3, e4 = feats

        def decoder(blk, inp, cin, tag, dropmul=None):
            mid = cin // 4
            a = conv_unit(tape, inp, blk.conv1.weight, blk.conv1.bias, seg(cin), stride=1, pad=0, bn=blk.abn1,
                          act=nv.ACT_LEAKY, slope=blk.abn1.slope, tag=tag + '.c1')
            b = conv_unit(tape, a, blk.deconv2.weight, blk.deconv2.bias, seg(mid), stride=2, pad=1, transposed=True,
                          bn=blk.abn2, act=nv.ACT_LEAKY, slope=blk.abn2.slope, tag=tag + '.dc')
            return conv_unit(tape, b, blk.conv3.weight, blk.conv3.bias, seg(mid), stride=1, pad=0, bn=blk.abn3,
                             act=nv.ACT_LEAKY, slope=blk.abn3.slope, dropmul=dropmul, tag=tag + '.c3')

        d4 = add(tape, decoder(self.decoder4, e4, 512, 'dec4'), e3, tag='skip3')
        d3 = add(tape, decoder(self.decoder3, d4, 256, 'dec3'), e2, tag='skip2')
        d2 = add(tape, decoder(self.decoder2, d3, 128, 'dec2'), e1, tag='skip1')
        drop = tape.dropout_table(tape.site('finaldrop1'), x.v.N, 64, self.finaldrop1.p)
        d1 = decoder(self.decoder1, d2, 64, 'dec1', dropmul=drop)           # Dropout2d(0.5) on d1, linknet.py:84
        lr1, lr2 = self.finalrelu1.negative_slope, self.finalrelu2.negative_slope
        f2 = conv_unit(tape, d1, self.finaldeconv1.weight, self.finaldeconv1.bias, seg(64), stride=2, pad=0,
                       transposed=True, act=nv.ACT_LEAKY, slope=lr1, tag='final.dc1')
        f4 = conv_unit(tape, f2, self.finalconv2.weight, self.finalconv2.bias, seg(32), stride=1, pad=0,
                       act=nv.ACT_LEAKY, slope=lr2, tag='final.c2')
        # finalconv3 (linknet.py:62: Conv2d(32, num_classes, 2, padding=1)): a classifier with a 2 x 2 window -- on the head kernels
        # where they serve it (one pass over f4 per direction; its backward also applies finalrelu2's mask), else as a convolution
        if nv.query('segnb_head_conv_ok', 32, self.num_classes, 2, 2):
            return head_conv(tape, f4, self.finalconv3.weight, self.finalconv3.bias, 1, dlogits, tag='final.c3')
        f5 = conv_unit(tape, f4, self.finalconv3.weight, self.finalconv3.bias, seg(32), stride=1, pad=1,
                       act=nv.ACT_NONE, tag='final.c3')
        return head_from_act(tape, f5, self.num_classes, dlogits)
