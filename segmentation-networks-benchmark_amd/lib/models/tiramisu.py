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

Channel layout: every tensor the kernels see has a multiple of 8 channels.  With a growth rate (or a first convolution)
that is not one -- FCDenseNet57, growth 12, tiramisu.py:187-191 -- each layer's slice of the concat buffer is padded to the
next multiple of 8 (12 -> 16, pad channels zero end to end): a prefix of the buffer is then a list of (real, padded) segments,
which the convolutions take as their input channel map and the per-channel BatchNorm as one launch set per segment
(segnb.net.bn_act).  With multiples of 8 (FCDenseNet67 / 103: growth 16, first convolution 48) every list collapses to one
contiguous segment and the launches are exactly those of the unpadded plan.
"""
import os

import torch
from torch import nn

from segnb import _native as nv
from segnb import convplan as cp
from segnb.engine import STAT_REPLICAS
from segnb.net import Act, HipNet, bn_act, conv_unit, head_1x1, head_from_act


def _merge(segs):
    """[(real, padded), ...] -> one contiguous segment when nothing is padded"""
    if all(r == p for r, p in segs):
        t = sum(r for r, _ in segs)
        return [(t, t)]
    return list(segs)


def _width(segs):
    return sum(p for _, p in segs)


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
    # Prefix statistics summed ONCE per slice (tiramisu.py:9-44: every DenseLayer's BatchNorm covers the whole concat prefix, whose
    # batch statistics do not change from layer to layer): one table per concat buffer, each slice's share accumulated by the pass
    # that writes it (segnb_bn_act_fwd_stats / segnb_bn_stats_ld), read as a channel range by every layer
    # (segnb_bn_fwd_fused_ld).  cache_prefix_stats = False (class attribute): a statistics pass over the prefix per layer (A/B)
    cache_prefix_stats = True

    def _dense_layers(self, tape, layers, buf, gbuf, off0, in_segs, tag, tbl=None):
        """Run DenseLayers in place inside buf (padded channel offset off0): layer l reads the prefix in_segs + l growth
        slices and writes the next slice (growth_rate real channels in pad8(growth_rate)).  tbl: (flat table tensor, element
        offset of the buffer's channel 0, row stride) of the buffer's statistics table, or None."""

        def src(ch):
            return None if tbl is None else (tbl[0], tbl[1] + ch, tbl[2])
        g = self.growth_rate
        gp = cp.pad8(g)
        for l, layer in enumerate(layers):
            segs = list(in_segs) + [(g, gp)] * l
            wl = _width(segs)
            prefix = Act(buf.slice(off0, wl))
            prefix.g = gbuf.slice(off0, wl) if gbuf is not None else None
            ms = _merge(segs)
            a = bn_act(tape, prefix, layer.norm, nv.ACT_RELU, tag=tag + '.norm', segs=ms if len(ms) > 1 else None,
                       stats_src=src(off0))
            drop = tape.dropout_table(tape.site(tag + '.drop'), buf.N, gp, layer.drop.p)
            out = conv_unit(tape, a, layer.conv.weight, layer.conv.bias, ms, act=nv.ACT_NONE, dropmul=drop,
                            out=buf.slice(off0 + wl, gp), tag=tag + '.conv', out_stats=src(off0 + wl))
            out.g = gbuf.slice(off0 + wl, gp) if gbuf is not None else None

    def _build(self, tape, x, dlogits):
        g, nd = self.growth_rate, len(self.down_blocks)
        gp = cp.pad8(g)
        N, H, W = x.v.N, x.v.H, x.v.W
        sizes = [(H, W)]
        for _ in range(nd):
            sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
        if min(sizes[-1]) < 1:
            raise ValueError('input too small for %d poolings' % nd)
        need = tape.need_grad
        # encoder block d: its input (real channels cin[d]) + down_blocks[d] growth slices
        cin = [self._first]
        for n in self.down_blocks:
            cin.append(cin[-1] + g * n)
        block_segs = [[(cin[d], cp.pad8(cin[d]))] + [(g, gp)] * self.down_blocks[d] for d in range(nd)]
        # decoder stage i pairs with encoder block d = nd-1-i; convT widths prev_i (real)
        prevs = [g * self.bottleneck_layers] + [g * n for n in self.up_blocks[:-1]]
        stage_segs = []
        ubufs, gbufs = [], []
        for i in range(nd):
            d = nd - 1 - i
            segs = [(prevs[i], cp.pad8(prevs[i]))] + block_segs[d] + [(g, gp)] * self.up_blocks[i]
            stage_segs.append(segs)
            ch = _width(segs)
            ubufs.append(tape.view('U%d' % i, N, sizes[d][0], sizes[d][1], ch))
            gb = tape.view('GU%d' % i, N, sizes[d][0], sizes[d][1], ch) if need else None
            if gb is not None:
                tape.rt.clear_view(gb)
            gbufs.append(gb)
        # ---- statistics tables of the concat buffers (stage buffers 0 .. nd-1, then the bottleneck buffer), cleared once per
        # step by Tape.begin()
        nb = self.bottleneck_layers
        bott_w = cp.pad8(cin[nd]) + gp * nb
        tbls = [None] * (nd + 1)
        if self.cache_prefix_stats and tape.fuses_finalize():
            widths = [_width(sg) for sg in stage_segs] + [bott_w]
            for k, wd in enumerate(widths):
                tbls[k] = (tape.step_zeroed('prefix_stats%d' % k, (STAT_REPLICAS, 2, wd), torch.float64), 0, wd)

        def src(k, ch):
            return None if tbls[k] is None else (tbls[k][0], tbls[k][1] + ch, tbls[k][2])
        # ---- encoder: block d lives in the skip slice of its decoder stage buffer
        inp, inp_seg = x, [(self._in_channels, x.v.Cp)]
        first = True
        for d in range(nd):
            i = nd - 1 - d
            U, G, off0 = ubufs[i], gbufs[i], cp.pad8(prevs[i])
            cur, curp = cin[d], cp.pad8(cin[d])
            if first:
                o = conv_unit(tape, inp, self.firstconv.weight, self.firstconv.bias, inp_seg, act=nv.ACT_NONE,
                              out=U.slice(off0, curp), tag='firstconv', out_stats=src(i, off0))
                first = False
            else:
                o = pooled_writer(U.slice(off0, curp), src(i, off0))
            o.g = G.slice(off0, curp) if G is not None else None
            self._dense_layers(tape, self.denseBlocksDown[d].layers, U, G, off0, block_segs[d][:1], 'down%d' % d, tbls[i])
            cur = cin[d + 1]
            wb = _width(block_segs[d])
            full = Act(U.slice(off0, wb))
            full.g = G.slice(off0, wb) if G is not None else None
            td = self.transDownBlocks[d]
            ms = _merge(block_segs[d])
            a = bn_act(tape, full, td.norm, nv.ACT_RELU, tag='td%d.norm' % d, segs=ms if len(ms) > 1 else None,
                       stats_src=src(i, off0))
            drop = tape.dropout_table(tape.site('td%d.drop' % d), N, cp.pad8(cur), td.drop.p)
            # conv1x1 -> dropout -> maxpool: the pooled tensor is the next block's input slice; a closure defers
            # the conv so that it can write its pooled output straight into the next buffer
            def pooled_writer(dst, st, a=a, td=td, ms=ms, drop=drop, d=d):
                _, p = conv_unit(tape, a, td.conv.weight, td.conv.bias, ms, stride=1, pad=0,
                                 act=nv.ACT_NONE, dropmul=drop, pool=True, pool_out=dst, tag='td%d.conv' % d, out_stats=st)
                return p
        # ---- bottleneck (its own buffer: pooled input + new layers; only the new layers go on)
        cur, curp = cin[nd], cp.pad8(cin[nd])
        B = tape.view('B', N, sizes[nd][0], sizes[nd][1], curp + gp * nb)
        GB = tape.view('GB', N, sizes[nd][0], sizes[nd][1], curp + gp * nb) if need else None
        if GB is not None:
            tape.rt.clear_view(GB)
        o = pooled_writer(B.slice(0, curp), src(nd, 0))
        o.g = GB.slice(0, curp) if GB is not None else None
        self._dense_layers(tape, self.bottleneck.bottleneck.layers, B, GB, 0, [(cur, curp)], 'bottleneck', tbls[nd])
        new, new_segs = Act(B.slice(curp, gp * nb)), [(g, gp)] * nb
        new.g = GB.slice(curp, gp * nb) if GB is not None else None
        # ---- decoder
        for i in range(nd):
            U, G = ubufs[i], gbufs[i]
            d = nd - 1 - i
            tu = self.transUpBlocks[i].convTrans
            cpv = cp.pad8(prevs[i])
            o = conv_unit(tape, new, tu.weight, tu.bias, _merge(new_segs), stride=2, pad=0, transposed=True, act=nv.ACT_NONE,
                          out=U.slice(0, cpv), out_hw=(U.H, U.W), tag='tu%d' % i, out_stats=src(i, 0))
            o.g = G.slice(0, cpv) if G is not None else None
            in_segs = [(prevs[i], cpv)] + block_segs[d]
            self._dense_layers(tape, self.denseBlocksUp[i].layers, U, G, 0, in_segs, 'up%d' % i, tbls[i])
            m = self.up_blocks[i]
            if i + 1 < nd:
                new, new_segs = Act(U.slice(_width(in_segs), gp * m)), [(g, gp)] * m
                new.g = G.slice(_width(in_segs), gp * m) if G is not None else None
            else:
                full = Act(U.slice(0, _width(stage_segs[i])))
                full.g = G.slice(0, _width(stage_segs[i])) if G is not None else None
                full_segs = _merge(stage_segs[i])
        if len(full_segs) == 1:
            return head_1x1(tape, full, self.finalConv.weight, self.finalConv.bias, dlogits)
        # padded slices inside the last buffer: the 1x1 classifier runs as a general convolution over the segment list
        o = conv_unit(tape, full, self.finalConv.weight, self.finalConv.bias, full_segs, stride=1, pad=0, act=nv.ACT_NONE,
                      tag='finalConv')
        return head_from_act(tape, o, self.num_classes, dlogits)


def FCDenseNet57(n_classes):
    return FCDenseNet(in_channels=3, down_blocks=(4, 4, 4, 4, 4), up_blocks=(4, 4, 4, 4, 4), bottleneck_layers=4,
                      growth_rate=12, out_chans_first_conv=48, n_classes=n_classes)


def FCDenseNet67(n_classes):
    return FCDenseNet(in_channels=3, down_blocks=(5, 5, 5, 5, 5), up_blocks=(5, 5, 5, 5, 5), bottleneck_layers=5,
                      growth_rate=16, out_chans_first_conv=48, n_classes=n_classes)


def FCDenseNet103(n_classes):
    return FCDenseNet(in_channels=3, down_blocks=(4, 5, 7, 10, 12), up_blocks=(12, 10, 7, 5, 4), bottleneck_layers=15,
                      growth_rate=16, out_chans_first_conv=48, n_classes=n_classes)
