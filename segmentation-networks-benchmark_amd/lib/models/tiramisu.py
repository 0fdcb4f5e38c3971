"""FCDenseNet ("One Hundred Layers Tiramisu") on the MI355X engine -- drop-in for the reference's
``lib.models.tiramisu`` (/root/reference/lib/models/tiramisu.py:9-205): ``FCDenseNet(in_channels, down_blocks,
up_blocks, bottleneck_layers, growth_rate, out_chans_first_conv, n_classes)`` and ``FCDenseNet57/67/103``, same
attribute tree / state_dict keys (``denseBlocksDown.i.layers.j.norm.weight`` ...), fp32 NCHW logits.

Memory plan: a dense block never copies.  Each decoder stage owns ONE buffer
``[ConvTranspose output | skip = the paired encoder block (input + its layers) | this stage's new layers]``;
the paired encoder block is *computed inside* the skip slice, every DenseLayer reads a channel-prefix view and
writes its growth_rate new channels into the next slice (torch.cat of tiramisu.py:36,43,72 costs nothing), and
center_crop (:86-90, offset always 0) is the iteration bound of the ConvTranspose launch.  Backward mirrors it with
one gradient buffer per stage buffer into which every consumer accumulates.

Pre-activation BatchNorm (norm -> relu -> conv, :12-15) gets its batch statistics from segnb_bn_stats.
Channel counts must be multiples of 8 (true for FCDenseNet67/103: growth 16, first conv 48).
"""
import torch
from torch import nn

from segnb import _native as nv
from segnb.net import Act, HipNet, bn_act, conv_unit, head_1x1


def _holder_forward(self, *a, **k):
    raise RuntimeError('parameter holder; run the whole FCDenseNet (HIP executor)')


class DenseLayer(nn.Sequential):
    def __init__(self, in_channels, growth_rate):
        super(DenseLayer, self).__init__()
        self.add_module('norm', nn.BatchNorm2d(in_channels))
        self.add_module('relu', nn.ReLU(True))
        self.add_module('conv', nn.Conv2d(in_channels, growth_rate, kernel_size=3, stride=1, padding=1, bias=True))
        self.add_module('drop', nn.Dropout2d(0.2))
    forward = _holder_forward


class DenseBlock(nn.Module):
    def __init__(self, in_channels, growth_rate, n_layers, upsample=False):
        super(DenseBlock, self).__init__()
        self.upsample = upsample
        self.layers = nn.ModuleList([DenseLayer(in_channels + i * growth_rate, growth_rate) for i in range(n_layers)])
    forward = _holder_forward


class TransitionDown(nn.Sequential):
    def __init__(self, in_channels):
        super(TransitionDown, self).__init__()
        self.add_module('norm', nn.BatchNorm2d(num_features=in_channels))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0, bias=True))
        self.add_module('drop', nn.Dropout2d(0.2))
        self.add_module('maxpool', nn.MaxPool2d(2))
    forward = _holder_forward


class TransitionUp(nn.Module):
    def __init__(self, in_channels, out_channels):
        super(TransitionUp, self).__init__()
        self.convTrans = nn.ConvTranspose2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=2,
                                            padding=0, bias=True)
    forward = _holder_forward


class Bottleneck(nn.Sequential):
    def __init__(self, in_channels, growth_rate, n_layers):
        super(Bottleneck, self).__init__()
        self.add_module('bottleneck', DenseBlock(in_channels, growth_rate, n_layers, upsample=True))
    forward = _holder_forward


class FCDenseNet(HipNet):
    def __init__(self, in_channels=3, down_blocks=(5, 5, 5, 5, 5), up_blocks=(5, 5, 5, 5, 5), bottleneck_layers=5,
                 growth_rate=16, out_chans_first_conv=48, n_classes=12):
        super(FCDenseNet, self).__init__()
        if growth_rate % 8 or out_chans_first_conv % 8:
            raise NotImplementedError('growth_rate and out_chans_first_conv must be multiples of 8 on this engine '
                                      '(FCDenseNet67 / 103 are; FCDenseNet57 with growth 12 is not wired by the '
                                      'reference either, torch_train.py:127-128)')
        if len(down_blocks) != len(up_blocks):
            raise ValueError('down_blocks and up_blocks must have the same length')
        self.num_classes = n_classes
        self.down_blocks, self.up_blocks = down_blocks, up_blocks
        self.growth_rate, self.bottleneck_layers = growth_rate, bottleneck_layers
        g = growth_rate
        skips = []
        self.add_module('firstconv', nn.Conv2d(in_channels=in_channels, out_channels=out_chans_first_conv,
                                               kernel_size=3, stride=1, padding=1, bias=True))
        cur = out_chans_first_conv
        self.denseBlocksDown = nn.ModuleList([])
        self.transDownBlocks = nn.ModuleList([])
        for n in down_blocks:
            self.denseBlocksDown.append(DenseBlock(cur, g, n))
            cur += g * n
            skips.insert(0, cur)
            self.transDownBlocks.append(TransitionDown(cur))
        self.add_module('bottleneck', Bottleneck(cur, g, bottleneck_layers))
        prev = g * bottleneck_layers
        cur += prev
        self.transUpBlocks = nn.ModuleList([])
        self.denseBlocksUp = nn.ModuleList([])
        for i in range(len(up_blocks) - 1):
            self.transUpBlocks.append(TransitionUp(prev, prev))
            cur = prev + skips[i]
            self.denseBlocksUp.append(DenseBlock(cur, g, up_blocks[i], upsample=True))
            prev = g * up_blocks[i]
            cur += prev
        self.transUpBlocks.append(TransitionUp(prev, prev))
        cur = prev + skips[-1]
        self.denseBlocksUp.append(DenseBlock(cur, g, up_blocks[-1], upsample=False))
        cur += g * up_blocks[-1]
        self.finalConv = nn.Conv2d(in_channels=cur, out_channels=n_classes, kernel_size=1, stride=1, padding=0,
                                   bias=True)
        self.softmax = nn.LogSoftmax(dim=1)          # defined but unused by the reference's forward (:166,183)
        self._first, self._skips = out_chans_first_conv, skips
        self._init_engine(in_channels)

    # ---- plan ------------------------------------------------------------------------------------------------
    def _dense_layers(self, tape, layers, buf, gbuf, off0, cin, tag):
        """Run DenseLayers in place inside buf: layer l reads channels [off0, off0+cin+l*g), writes the next g."""
        g = self.growth_rate
        for l, layer in enumerate(layers):
            cl = cin + l * g
            prefix = Act(buf.slice(off0, cl))
            prefix.g = gbuf.slice(off0, cl) if gbuf is not None else None
            a = bn_act(tape, prefix, layer.norm, nv.ACT_RELU, tag=tag + '.norm')
            drop = tape.dropout_table(tape.site(tag + '.drop'), buf.N, g, layer.drop.p)
            out = conv_unit(tape, a, layer.conv.weight, layer.conv.bias, [(cl, cl)], act=nv.ACT_NONE, dropmul=drop,
                            out=buf.slice(off0 + cl, g), tag=tag + '.conv')
            out.g = gbuf.slice(off0 + cl, g) if gbuf is not None else None

    def _build(self, tape, x, dlogits):
        g, nd = self.growth_rate, len(self.down_blocks)
        N, H, W = x.v.N, x.v.H, x.v.W
        sizes = [(H, W)]
        for _ in range(nd):
            sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
        if min(sizes[-1]) < 1:
            raise ValueError('input too small for %d poolings' % nd)
        need = tape.need_grad
        # decoder stage i pairs with encoder block d = nd-1-i; convT widths prev_i
        prevs = [g * self.bottleneck_layers] + [g * n for n in self.up_blocks[:-1]]
        ubufs, gbufs = [], []
        for i in range(nd):
            d = nd - 1 - i
            ch = prevs[i] + self._skips[i] + g * self.up_blocks[i]
            ubufs.append(tape.view('U%d' % i, N, sizes[d][0], sizes[d][1], ch))
            gb = tape.view('GU%d' % i, N, sizes[d][0], sizes[d][1], ch) if need else None
            if gb is not None:
                tape.rt.clear_view(gb)
            gbufs.append(gb)
        # ---- encoder: block d lives in the skip slice of its decoder stage buffer
        cur = self._first
        inp, inp_seg = x, [(self._in_channels, x.v.Cp)]
        first = True
        for d in range(nd):
            i = nd - 1 - d
            U, G, off0 = ubufs[i], gbufs[i], prevs[i]
            if first:
                o = conv_unit(tape, inp, self.firstconv.weight, self.firstconv.bias, inp_seg, act=nv.ACT_NONE,
                              out=U.slice(off0, cur), tag='firstconv')
                first = False
            else:
                o = pooled_writer(U.slice(off0, cur))
            o.g = G.slice(off0, cur) if G is not None else None
            self._dense_layers(tape, self.denseBlocksDown[d].layers, U, G, off0, cur, 'down%d' % d)
            cur += g * self.down_blocks[d]
            full = Act(U.slice(off0, cur))
            full.g = G.slice(off0, cur) if G is not None else None
            td = self.transDownBlocks[d]
            a = bn_act(tape, full, td.norm, nv.ACT_RELU, tag='td%d.norm' % d)
            drop = tape.dropout_table(tape.site('td%d.drop' % d), N, cur, td.drop.p)
            # conv1x1 -> dropout -> maxpool: the pooled tensor is the next block's input slice; a closure defers
            # the conv so that it can write its pooled output straight into the next buffer
            def pooled_writer(dst, a=a, td=td, cur=cur, drop=drop, d=d):
                _, p = conv_unit(tape, a, td.conv.weight, td.conv.bias, [(cur, cur)], stride=1, pad=0,
                                 act=nv.ACT_NONE, dropmul=drop, pool=True, pool_out=dst, tag='td%d.conv' % d)
                return p
        # ---- bottleneck (its own buffer: pooled input + new layers; only the new layers go on)
        nb = self.bottleneck_layers
        B = tape.view('B', N, sizes[nd][0], sizes[nd][1], cur + g * nb)
        GB = tape.view('GB', N, sizes[nd][0], sizes[nd][1], cur + g * nb) if need else None
        if GB is not None:
            tape.rt.clear_view(GB)
        o = pooled_writer(B.slice(0, cur))
        o.g = GB.slice(0, cur) if GB is not None else None
        self._dense_layers(tape, self.bottleneck.bottleneck.layers, B, GB, 0, cur, 'bottleneck')
        new = Act(B.slice(cur, g * nb))
        new.g = GB.slice(cur, g * nb) if GB is not None else None
        # ---- decoder
        for i in range(nd):
            U, G = ubufs[i], gbufs[i]
            tu = self.transUpBlocks[i].convTrans
            c = prevs[i]
            o = conv_unit(tape, new, tu.weight, tu.bias, [(c, c)], stride=2, pad=0, transposed=True, act=nv.ACT_NONE,
                          out=U.slice(0, c), out_hw=(U.H, U.W), tag='tu%d' % i)
            o.g = G.slice(0, c) if G is not None else None
            cin = c + self._skips[i]
            self._dense_layers(tape, self.denseBlocksUp[i].layers, U, G, 0, cin, 'up%d' % i)
            nn_ = g * self.up_blocks[i]
            if i + 1 < nd:
                new = Act(U.slice(cin, nn_))
                new.g = G.slice(cin, nn_) if G is not None else None
            else:
                full = Act(U.slice(0, cin + nn_))
                full.g = G.slice(0, cin + nn_) if G is not None else None
        return head_1x1(tape, full, self.finalConv.weight, self.finalConv.bias, dlogits)


def FCDenseNet57(n_classes):
    return FCDenseNet(in_channels=3, down_blocks=(4, 4, 4, 4, 4), up_blocks=(4, 4, 4, 4, 4), bottleneck_layers=4,
                      growth_rate=12, out_chans_first_conv=48, n_classes=n_classes)


def FCDenseNet67(n_classes):
    return FCDenseNet(in_channels=3, down_blocks=(5, 5, 5, 5, 5), up_blocks=(5, 5, 5, 5, 5), bottleneck_layers=5,
                      growth_rate=16, out_chans_first_conv=48, n_classes=n_classes)


def FCDenseNet103(n_classes):
    return FCDenseNet(in_channels=3, down_blocks=(4, 5, 7, 10, 12), up_blocks=(12, 10, 7, 5, 4), bottleneck_layers=15,
                      growth_rate=16, out_chans_first_conv=48, n_classes=n_classes)
