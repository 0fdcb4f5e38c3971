"""TernausNet-16 (VGG16 encoder U-Net) on the MI355X engine -- drop-in for the reference's ``lib.models.unet16``
(/root/reference/lib/models/unet16.py:12-131): ``UNet16(num_classes=1, num_filters=32, pretrained=False)``, same
attribute tree / state_dict keys (``encoder.N.*`` aliased by ``convK.M.*``, ``center.block.0.conv.weight``,
``dec5.block.1.weight`` ...), fp32 NCHW logits.

The reference takes the encoder from ``torchvision.models.vgg16(...).features``; torchvision is not a dependency
here, so the same 13-conv stack (cfg "D", conv indices 0,2,5,7,10,12,14,17,19,21,24,26,28 -- unet16.py:73-102) is
built locally with torchvision's initialisation (kaiming_normal fan_out, zero bias).  ``pretrained='vgg'`` needs
downloaded weights: pass a state_dict instead (load_state_dict works with the reference's checkpoints).

Forward/backward run on the define-by-run executor (segnb.net): conv+ReLU units with fused MaxPool2d(2),
ConvTranspose2d(4, stride 2, pad 1) as four output-parity gather launches, zero-copy concatenation.
"""
import os

import torch
from torch import nn

from segnb import _native as nv
from segnb import convplan as cp
from segnb.net import HipNet, concat, conv_unit, head_1x1, upsample_bilinear2x

VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def vgg16_features():
    layers, cin = [], 3
    for v in VGG16_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = nn.Conv2d(cin, v, kernel_size=3, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(conv.bias, 0)
            layers += [conv, nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class ConvRelu(nn.Module):
    def __init__(self, in_, out):
        super(ConvRelu, self).__init__()
        self.conv = nn.Conv2d(in_, out, 3, padding=1)
        self.activation = nn.ReLU(inplace=True)

    def forward(self, x):
        raise RuntimeError('parameter holder; run the whole UNet16 (HIP executor)')


class DecoderBlock(nn.Module):
    """unet16.py:24-49: conv3x3 + ReLU -> ConvTranspose2d(4, 2, 1) -> ReLU (is_deconv, what UNet16 builds), or
    Upsample(x2, bilinear) -> conv3x3 + ReLU -> conv3x3 + ReLU; same attribute tree / state_dict keys either way."""

    def __init__(self, in_channels, middle_channels, out_channels, is_deconv=True):
        super(DecoderBlock, self).__init__()
        self.in_channels = in_channels
        self.is_deconv = bool(is_deconv)
        if is_deconv:
            self.block = nn.Sequential(ConvRelu(in_channels, middle_channels),
                                       nn.ConvTranspose2d(middle_channels, out_channels, kernel_size=4, stride=2, padding=1),
                                       nn.ReLU(inplace=True))
        else:
            self.block = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear'),
                                       ConvRelu(in_channels, middle_channels),
                                       ConvRelu(middle_channels, out_channels))

    def forward(self, x):
        raise RuntimeError('parameter holder; run the whole UNet16 (HIP executor)')


class UNet16(HipNet):
    # The weight gradients of this net (no BatchNorm passes on the dependent chain, 64-512 channel convolutions at 1024^2 .. 64^2) are
    # the LONGER of the two streams: at the library's default share (half of the CUs) the chain ended 1.7 ms before them and
    # waited (profiles/r04_ab.txt: 18.75 ms per step at 50 %, 17.98 at 75 %, 17.77 at 100 %; LinkNet34 loses 7 % at 100 %).
    wg_cu_pct = 100

    def __init__(self, num_classes=1, num_filters=32, pretrained=False, is_deconv=True):
        """is_deconv: the DecoderBlock branch (unet16.py:34-46).  The reference's UNet16 always builds the deconvolution branch
        (:104-108 pass no flag); False selects its class's other, bilinear-upsampling branch for every decoder block."""
        super(UNet16, self).__init__()
        if pretrained == 'vgg':
            raise ValueError("pretrained='vgg' needs downloaded torchvision weights; load a state_dict instead")
        self.num_classes = num_classes
        self.pool = nn.MaxPool2d(2, 2)
        self.encoder = vgg16_features()
        self.relu = nn.ReLU(inplace=True)
        e = self.encoder
        self.conv1 = nn.Sequential(e[0], self.relu, e[2], self.relu)
        self.conv2 = nn.Sequential(e[5], self.relu, e[7], self.relu)
        self.conv3 = nn.Sequential(e[10], self.relu, e[12], self.relu, e[14], self.relu)
        self.conv4 = nn.Sequential(e[17], self.relu, e[19], self.relu, e[21], self.relu)
        self.conv5 = nn.Sequential(e[24], self.relu, e[26], self.relu, e[28], self.relu)
        nf = num_filters
        self.center = DecoderBlock(512, nf * 8 * 2, nf * 8, is_deconv)
        self.dec5 = DecoderBlock(512 + nf * 8, nf * 8 * 2, nf * 8, is_deconv)
        self.dec4 = DecoderBlock(512 + nf * 8, nf * 8 * 2, nf * 8, is_deconv)
        self.dec3 = DecoderBlock(256 + nf * 8, nf * 4 * 2, nf * 2, is_deconv)
        self.dec2 = DecoderBlock(128 + nf * 2, nf * 2 * 2, nf, is_deconv)
        self.dec1 = ConvRelu(64 + nf, nf)
        self.final = nn.Conv2d(nf, num_classes, kernel_size=1)
        self._nf = nf
        self._init_engine(3)

    def _check_input(self, x):
        super(UNet16, self)._check_input(x)
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError('UNet16 needs H and W divisible by 32, got %dx%d' % (x.shape[2], x.shape[3]))

    def _build(self, tape, x, dlogits):
        nf, N, H, W = self._nf, x.v.N, x.v.H, x.v.W
        enc_idx = [[0, 2], [5, 7], [10, 12, 14], [17, 19, 21], [24, 26, 28]]
        enc_c = [64, 128, 256, 512, 512]
        dec_c = [nf, nf * 2, nf * 8, nf * 8, nf * 8]          # channels of the decoder tensor concatenated at level k
        # concat buffers [decoder tensor | encoder skip] at levels 1..5 (unet16.py:122-127)
        cats = []
        for k in range(5):
            cats.append(tape.view('cat%d' % k, N, H >> k, W >> k, cp.pad8(dec_c[k]) + cp.pad8(enc_c[k])))
        h, cin, skips = x, 3, []
        for k in range(5):
            convs = [self.encoder[i] for i in enc_idx[k]]
            for j, conv in enumerate(convs):
                last = j == len(convs) - 1
                segs = [(cin, cp.pad8(cin))]
                if last:
                    skip, h = conv_unit(tape, h, conv.weight, conv.bias, segs, act=nv.ACT_RELU, pool=True,
                                        out=cats[k].slice(cp.pad8(dec_c[k]), cp.pad8(enc_c[k])), tag='enc%d' % k)
                    skips.append(skip)
                else:
                    h = conv_unit(tape, h, conv.weight, conv.bias, segs, act=nv.ACT_RELU, tag='enc%d' % k)
                cin = conv.out_channels

        def decoder_block(blk, inp, segs, out_view, tag):
            if not blk.is_deconv:
                # Upsample(x2, bilinear) -> ConvRelu -> ConvRelu (unet16.py:42-46).  A concat input is upsampled as one tensor
                c1, c2 = blk.block[1].conv, blk.block[2].conv
                up = upsample_bilinear2x(tape, inp, tag=tag + '.up')
                mid = conv_unit(tape, up, c1.weight, c1.bias, segs, act=nv.ACT_RELU, tag=tag + '.conv1')
                return conv_unit(tape, mid, c2.weight, c2.bias, [(c1.out_channels, cp.pad8(c1.out_channels))], act=nv.ACT_RELU,
                                 out=out_view, tag=tag + '.conv2')
            c0, ct = blk.block[0].conv, blk.block[1]
            mid = conv_unit(tape, inp, c0.weight, c0.bias, segs, act=nv.ACT_RELU, tag=tag + '.conv')
            return conv_unit(tape, mid, ct.weight, ct.bias, [(c0.out_channels, cp.pad8(c0.out_channels))], stride=2,
                             pad=1, transposed=True, act=nv.ACT_RELU, out=out_view, tag=tag + '.deconv')

        d = decoder_block(self.center, h, [(512, 512)], cats[4].slice(0, cp.pad8(dec_c[4])), 'center')
        for k, blk in zip((4, 3, 2, 1), (self.dec5, self.dec4, self.dec3, self.dec2)):
            cat = concat(tape, [(d, 0), (skips[k], cp.pad8(dec_c[k]))], cats[k])
            segs = [(dec_c[k], cp.pad8(dec_c[k])), (enc_c[k], cp.pad8(enc_c[k]))]
            d = decoder_block(blk, cat, segs, cats[k - 1].slice(0, cp.pad8(dec_c[k - 1])), 'dec%d' % (k + 1))
        cat = concat(tape, [(d, 0), (skips[0], cp.pad8(dec_c[0]))], cats[0])
        d1 = conv_unit(tape, cat, self.dec1.conv.weight, self.dec1.conv.bias,
                       [(dec_c[0], cp.pad8(dec_c[0])), (enc_c[0], cp.pad8(enc_c[0]))], act=nv.ACT_RELU, tag='dec1')
        return head_1x1(tape, d1, self.final.weight, self.final.bias, dlogits)
