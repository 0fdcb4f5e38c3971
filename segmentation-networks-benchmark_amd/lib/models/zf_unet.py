"""ZF_UNET on the MI355X engine -- drop-in for the reference's ``lib.models.zf_unet``.

Same public surface as /root/reference/lib/models/zf_unet.py:35-95: constructor signature
``ZF_UNET(dropout_val=0.2, batch_norm=True, input_channels=3, num_classes=1, filters=32)``, attribute
tree / ``state_dict`` keys (``conv_224.l1.conv.weight`` ... ``conv_final.bias``, 156 entries at the
defaults), ``num_classes``, fp32 ``[N,3,H,W] -> [N,num_classes,H,W]`` logits, autograd through to
every parameter.  Parameters are ordinary ``nn.Parameter``s created by ``nn.Conv2d`` /
``nn.BatchNorm2d`` in the reference's construction order, so default initialisation under a given
``torch.manual_seed`` matches the reference's bit for bit.

Nothing here calls torch operators on the hot path: ``forward`` hands the batch to a static plan of
libsegnb_hip.so launches (segnb.engine) -- implicit-GEMM MFMA convolutions with BatchNorm statistics in
the epilogue, one fused BN+ReLU+Dropout2d+MaxPool/Upsample pass per conv writing straight into the
decoder's concat buffers (no torch.cat), hand-written backward.
"""
import os

import torch
from torch import nn

from segnb import _native as nv
from segnb import convplan as cp
from segnb.engine import ConvOp, FlatParams, InputNorm, PackTable, Runtime, Stage, UpCatConvOp, View, pack_input

ENCODER = ('conv_224', 'conv_112', 'conv_56', 'conv_28', 'conv_14', 'conv_7')
DECODER = ('up_conv_14', 'up_conv_28', 'up_conv_56', 'up_conv_112', 'up_conv_224')


class _ConvUnit(nn.Module):
    """Parameter holder named like the reference's _Conv3BN (zf_unet.py:5-10): .conv, .bn, .activation"""

    def __init__(self, cin, cout, bn):
        super(_ConvUnit, self).__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)
        self.bn = nn.BatchNorm2d(cout) if bn else None
        self.activation = nn.ReLU(inplace=True)

    def forward(self, x):
        raise RuntimeError('sub-blocks are parameter holders; run the whole ZF_UNET (fused HIP plan)')


class _Block(nn.Module):
    """Parameter holder named like _DoubleConvModule (zf_unet.py:20-25): .l1, .l2, .dropout"""

    def __init__(self, cin, cout, dropout_val, bn):
        super(_Block, self).__init__()
        self.l1 = _ConvUnit(cin, cout, bn)
        self.l2 = _ConvUnit(cout, cout, bn)
        self.dropout = nn.Dropout2d(p=dropout_val)

    def forward(self, x):
        raise RuntimeError('sub-blocks are parameter holders; run the whole ZF_UNET (fused HIP plan)')


class ZF_UNET(nn.Module):
    def __init__(self, dropout_val=0.2, batch_norm=True, input_channels=3, num_classes=1, filters=32):
        super(ZF_UNET, self).__init__()
        self.num_classes = num_classes
        self.pool = nn.MaxPool2d(2)
        self.unpool = nn.Upsample(scale_factor=2)
        f = filters
        widths = [f, 2 * f, 4 * f, 8 * f, 16 * f, 32 * f]
        cin = input_channels
        for name, w in zip(ENCODER, widths):
            setattr(self, name, _Block(cin, w, dropout_val, batch_norm))
            cin = w
        for name, lvl in zip(DECODER, (4, 3, 2, 1, 0)):
            setattr(self, name, _Block(widths[lvl + 1] + widths[lvl], widths[lvl], dropout_val, batch_norm))
        self.conv_final = nn.Conv2d(f, num_classes, 1)
        self._cfg = dict(dropout=float(dropout_val), bn=bool(batch_norm), cin=input_channels, widths=widths)
        # engine state (not part of state_dict)
        self.compute_dtype = 'bf16'     # 'bf16' = throughput path, 'f32' = exact-fp32 MFMA parity path
        self._engine = None
        self.dropout_override = None    # {block name: fp32 [N, C] multiplier table} -- Dropout2d replay
        self.input_norm = InputNorm()   # applied on the device when forward() is given a uint8 NHWC batch

    def set_compute_dtype(self, dtype):
        if dtype not in ('bf16', 'f32'):
            raise ValueError("compute dtype must be 'bf16' or 'f32'")
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._engine = None
        return self

    def _get_engine(self, device):
        e = self._engine
        if e is None or e.rt.device != device:
            e = _ZFUnetPlan(self, device)
            self._engine = e
        return e

    def forward(self, x):
        """x: float32 [N, C, H, W] (already normalised: what torch_train.py:177 hands the model), or the batch as the
        dataset holds it -- uint8 [N, H, W, C] -- which is normalised with ``self.input_norm`` inside the first
        convolution (lib/augmentations.py:452-460 + lib/common.py:70 fused into the kernel)."""
        u8 = x.dtype == torch.uint8
        cdim, hdim, wdim = (3, 1, 2) if u8 else (1, 2, 3)
        if x.dim() != 4 or x.shape[cdim] != self._cfg['cin']:
            raise ValueError('expected input [N, %d, H, W] (float) or [N, H, W, %d] (uint8), got %s %s'
                             % (self._cfg['cin'], self._cfg['cin'], x.dtype, tuple(x.shape)))
        if x.shape[hdim] % 32 or x.shape[wdim] % 32:
            raise ValueError('ZF_UNET needs H and W divisible by 32 (five 2x poolings), got %dx%d'
                             % (x.shape[hdim], x.shape[wdim]))
        eng = self._get_engine(x.device)
        x = x.detach().contiguous() if u8 else x.detach().contiguous().float()
        if torch.is_grad_enabled():
            # ONE parameter ties the output to the autograd graph (the backward plan writes every .grad itself, as views
            # of the flat gradient buffer, and returns no gradients): 70 fewer edges for autograd to set up and check
            anchor = next((p for p in eng.flat.param_list() if p.requires_grad), None)
            if anchor is not None:
                return _ZFUnetFn.apply(eng, x, anchor)
        return eng.forward(x, self.training, False)


class _ZFUnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, anchor):
        ctx.eng = eng
        out = eng.forward(x, eng.module.training, True)
        ctx.generation = eng.generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.generation != ctx.eng.generation:
            # one set of activation / dropout / statistics buffers per input geometry: a later forward has overwritten
            # what this graph's backward needs -- refuse instead of using the wrong activations
            raise RuntimeError('ZF_UNET: another forward ran on this model since the forward being differentiated; '
                               'run backward before the next forward (the plan keeps ONE set of activation buffers)')
        ctx.eng.backward(dlogits.contiguous().float())
        return None, None, None


class _ZFUnetPlan(object):
    """Static launch plan of one ZF_UNET on one device."""

    def __init__(self, module, device):
        self.module = module
        self.rt = rt = Runtime(device, module.compute_dtype)
        self.flat = FlatParams(module)
        self.flat.ensure(rt.device)
        cfg = module._cfg
        widths = cfg['widths']
        self.widths = widths
        self.wp = [cp.pad8(w) for w in widths]
        self.cin_p = cp.pad8(cfg['cin'])
        self.p_drop = cfg['dropout']
        self.stages = {}
        for i, name in enumerate(ENCODER):
            blk = getattr(module, name)
            seg1 = [(cfg['cin'], self.cin_p)] if i == 0 else [(widths[i - 1], self.wp[i - 1])]
            self._add(name, blk, seg1, need_dgrad_l1=(i > 0))
        # Decoder blocks: the backward of conv3x3(cat([Upsample x2(u), skip])) by input segment, the upsampled one on the
        # low-resolution tensor (segnb.engine.UpCatConvOp: 4 x 4 / stride-2 transposed-convolution identity, 16 instead of 36
        # multiply-adds per low-resolution pixel).  Throughput path only: the exact-fp32 parity mode keeps the plain plan.
        sp = os.environ.get('SEGNB_SUBPIXEL', 'auto')
        self.subpixel = (module.compute_dtype == 'bf16') if sp == 'auto' else sp not in ('0', 'off')
        for name, lvl in zip(DECODER, (4, 3, 2, 1, 0)):
            blk = getattr(module, name)
            seg1 = [(widths[lvl + 1], self.wp[lvl + 1]), (widths[lvl], self.wp[lvl])]
            self._add(name, blk, seg1, True, upcat=self.subpixel)
        from segnb.engine import ReplayGuard
        self._guard_f, self._guard_b = ReplayGuard('ZF_UNET forward'), ReplayGuard('ZF_UNET backward')
        self._guard_mode = (None, None)
        self._bufs = {}
        self._pack_tables = {}
        self._retired_tables = []
        self._packed_key = None
        self._cplans = {}
        self.generation = 0
        self.K = module.num_classes

    def _add(self, name, blk, seg1, need_dgrad_l1, upcat=False):
        rt = self.rt
        cout = blk.l1.conv.out_channels
        if upcat:
            c1 = UpCatConvOp(rt, blk.l1.conv.weight, blk.l1.conv.bias, seg1, need_dgrad_l1)
        else:
            c1 = ConvOp(rt, blk.l1.conv.weight, blk.l1.conv.bias, seg1, 1, 1, False, need_dgrad_l1)
        c2 = ConvOp(rt, blk.l2.conv.weight, blk.l2.conv.bias, [(cout, cp.pad8(cout))], 1, 1, False, True)
        self.stages[name] = (Stage(rt, c1, blk.l1.bn, nv.ACT_RELU, 0.0, name + '.l1'),
                             Stage(rt, c2, blk.l2.bn, nv.ACT_RELU, 0.0, name + '.l2'))
        for st in self.stages[name]:
            st.defer_unpack = True         # one batched unpack at the end of backward()

    # ---- buffers for one input geometry -------------------------------------------------------------
    def buffers(self, N, H, W):
        key = (N, H, W)
        b = self._bufs.get(key)
        if b is not None:
            return b
        rt, wp = self.rt, self.wp
        hs = [(H >> i, W >> i) for i in range(6)]
        b = {'x': View.alloc(rt, N, H, W, self.cin_p)}
        for i in range(6):
            h, w = hs[i]
            b['a1_%d' % i] = View.alloc(rt, N, h, w, wp[i])          # encoder l1 activated
            b['da1_%d' % i] = View.alloc(rt, N, h, w, wp[i])
            if i > 0:
                b['p_%d' % i] = View.alloc(rt, N, h, w, wp[i - 1])   # pooled input of level i
                b['dp_%d' % i] = View.alloc(rt, N, h, w, wp[i - 1])
            if i < 5:
                b['cat_%d' % i] = View.alloc(rt, N, h, w, wp[i + 1] + wp[i])   # [upsampled | skip]
                b['dcat_%d' % i] = View.alloc(rt, N, h, w, wp[i + 1] + wp[i])
                b['b1_%d' % i] = View.alloc(rt, N, h, w, wp[i])      # decoder l1 activated
                b['db1_%d' % i] = View.alloc(rt, N, h, w, wp[i])
                if self.subpixel:
                    # the tensor cat_i's first segment is upsampled from, at its own (half) resolution, and its gradient
                    h2, w2 = hs[i + 1]
                    if self._low_res(i, N, H, W):
                        b['u_%d' % i] = View.alloc(rt, N, h2, w2, wp[i + 1])
                    if self._seg(i, N, H, W):
                        b['du_%d' % i] = View.alloc(rt, N, h2, w2, wp[i + 1])
        b['f0'] = View.alloc(rt, N, H, W, wp[0])
        b['df0'] = View.alloc(rt, N, H, W, wp[0])
        b['dlogits_in'] = torch.zeros((N, self.K, H, W), dtype=torch.float32, device=rt.device)
        sizes = [(n, self.stages[n][1].Cp) for n in ENCODER + DECODER]
        b['drop_flat'] = torch.ones(sum(N * c for _, c in sizes), dtype=torch.float32, device=rt.device)
        b['drop'], off = {}, 0
        for n, c in sizes:
            b['drop'][n] = b['drop_flat'][off:off + N * c].view(N, c)
            off += N * c
        self._bufs[key] = b
        return b

    def _conv_sizes(self, H, W):
        """(ConvOp, input height, input width) of every convolution of the net."""
        out = []
        for i, name in enumerate(ENCODER):
            out += [(st.conv, H >> i, W >> i) for st in self.stages[name]]
        for name, lvl in zip(DECODER, (4, 3, 2, 1, 0)):
            out += [(st.conv, H >> lvl, W >> lvl) for st in self.stages[name]]
        return out

    # gradient groups in the order backward finishes them = from the END of the flat parameter buffer: the decoder
    # (+ head), the two deepest encoder blocks (71 MB of the 126 MB of gradients), the rest.  Each group is unpacked
    # as soon as its weight gradients exist (on the side stream, beside the remaining backward) and handed to the
    # data-parallel hook, so its all-reduce overlaps the rest of backward.
    # (sizing the persistent data-gradient grids for fewer CUs while the weight gradients hold half of the chip was measured in
    # round 1: 100 % -> 5.33 ms/step, 75 % -> 5.43, 50 % -> 5.45; the knob is gone)
    UNPACK_GROUPS = ((2 * len(ENCODER), None), (8, 2 * len(ENCODER)), (0, 8))       # conv index ranges

    def _seg(self, lvl, N, H, W):
        """Does decoder level lvl run its first convolution's data gradient by segment at this input size?"""
        conv = self.stages[DECODER[4 - lvl]][0].conv
        return self.subpixel and conv.writes_du(N, H >> lvl, W >> lvl)

    def _low_res(self, lvl, N, H, W):
        """Does decoder level lvl read the tensor its concat buffer's first segment is upsampled from (segmented forward or
        weight gradient)?  -> the producer also writes it at its own resolution"""
        conv = self.stages[DECODER[4 - lvl]][0].conv
        return self.subpixel and conv.needs_low_res(N, H >> lvl, W >> lvl)

    def _upsampled(self, lvl, N, H, W):
        """Is the upsampled copy in the concat buffer of level lvl still read (9-tap forward or weight gradient)?"""
        conv = self.stages[DECODER[4 - lvl]][0].conv
        return not self.subpixel or conv.reads_upsampled(N, H >> lvl, W >> lvl)

    def _tables(self, H, W, N=None):
        """One-launch weight pack table and per-group gradient unpack tables for this input size (rebuilt if the
        flat parameter buffers were re-created).  -> (key, pack, (unpack_dec, unpack_deep, unpack_rest), (lo, lo))"""
        N = self._last_N if N is None else N
        self._last_N = N
        key = (N, H, W, self.flat.flat_p.data_ptr(), self.flat.flat_g.data_ptr())
        # keyed by the FULL geometry: a recorded backward list / HIP graph holds the raw device pointer of its unpack job
        # tables, so a forward at the same H x W with another batch size (validation, a partial last batch) must not
        # replace -- and thereby free -- the tables a list recorded for the first batch size still points at
        t = self._pack_tables.get((N, H, W))
        if t is None or t[0] != key:
            if t is not None:
                self._retired_tables.append(t)       # (lists recorded against the old flat buffers keep valid pointers)
            convs = self._conv_sizes(H, W)
            pj = []
            for conv, h, w in convs:
                pj += conv.pack_jobs(h, w, N) if isinstance(conv, UpCatConvOp) else conv.pack_jobs(h, w)
            unpacks, los = [], []
            for a, b in self.UNPACK_GROUPS:
                uj = []
                for conv, h, w in convs[a:b]:
                    uj += conv.unpack_jobs(h, w, self.flat.grad_of(conv.weight))
                unpacks.append(PackTable(self.rt, uj, 'segnb_unpack_wgrad_multi', 'segnb_unpack_wgrad'))
                los.append(self.flat._off[id(convs[a][0].weight)][0])       # first flat offset of the group
            # main stream: ONE launch of pair jobs (both matrices of a layer from one read of its parameter) + one of what
            # they do not take.  The data-gradient matrices among the latter (the segmented decoder levels: masked jobs with
            # few, slow blocks, 20 us of the step's start) are not needed before the backward: side stream, when there is one
            defer = (lambda j: j.get('form') == 'd') if self.rt.side_stream() is not None else None
            main = PackTable(self.rt, pj, 'segnb_pack_weight_multi', 'segnb_pack_weight', defer=defer)
            late = PackTable(self.rt, main.deferred, 'segnb_pack_weight_multi', 'segnb_pack_weight') if main.deferred else None
            packs = (main, late)
            t = (key, packs, tuple(unpacks), tuple(los))
            self._pack_tables[(N, H, W)] = t
        return t

    def _unpack_group(self, H, W, gi):
        """Unpack gradient group gi behind its weight gradients (side stream when there is one) and tell the
        data-parallel hook that everything from its first flat offset on is final."""
        rt = self.rt
        _, _, unpacks, los = self._tables(H, W)
        side = rt.side_stream() if getattr(rt, '_side_busy', False) else None
        if side is not None:
            with torch.cuda.stream(side):
                unpacks[gi].run()
        else:
            unpacks[gi].run()
        hook = getattr(self.module, '_grad_ready_hook', None)
        if hook is not None and gi < 2:
            # a recorded launch list is cut where the hook runs, and the hook itself runs OUTSIDE the recording: what it
            # launches (the optimizer update of a bucket behind its all-reduce, DataParallel.fuse_optimizer) is issued by the
            # live hook of every step -- recorded as well, a replay applied those updates twice
            self._plan_cut(('ready', gi, side is not None), resume=False)
            try:
                hook(self.flat, los[gi], (side,) if side is not None else ())
            finally:
                self._plan_resume()

    _last_N = None

    def _pack_key(self, H, W, N):
        return (sum(p._version for p in self.flat.param_list()), self.flat.version, N, H, W, self.flat.flat_p.data_ptr())

    # ---- optimizer.step() of plain SGD fused with this pack (segnb.optim.SGD -> FlatParams.sgd_pack_hook) --------------------------
    _pairs_key = None          # the pack key under which the PAIRED matrices are already those of the current parameters
    _sgd_geom = None

    def _sgd_pack(self, lr):
        """w -= lr * g on every parameter: the convolution weights of the pair-job table inside the pack kernel (their packed
        matrices for the geometry of the last differentiated forward leave the same pass), everything else by segnb_sgd_ranges."""
        if self._sgd_geom is None or self.rt.code != nv.BF16 or self.rt.device.type != 'cuda':
            return False
        N, H, W = self._sgd_geom
        t = self._pack_tables.get((N, H, W))
        if t is None or t[0] != (N, H, W, self.flat.flat_p.data_ptr(), self.flat.flat_g.data_ptr()):
            return False
        main = t[1][0]
        if main.ptable is None:
            return False
        comp = getattr(main, '_sgd_complement', None)
        if comp is None:
            base, total = self.flat.flat_p.data_ptr(), self.flat.total
            covered = sorted(((a - base) // 4, n) for a, n in main.pair_params)
            rows, pos, acc = [], 0, 0
            for start, n in covered + [(total, 0)]:
                if start > pos:
                    rows.append((pos, start - pos, acc))
                    acc += start - pos
                pos = max(pos, start + n)
            tab = torch.tensor(rows, dtype=torch.int64, device=self.rt.device) if rows else None
            comp = main._sgd_complement = (tab, len(rows), acc)
        main.run_pairs_sgd(self.flat.flat_p, self.flat.flat_g, lr)
        if comp[1]:
            nv.call('segnb_sgd_ranges', nv.ptr(self.flat.flat_p), nv.ptr(self.flat.flat_g), nv.ptr(comp[0]), comp[1], comp[2],
                    float(lr), self.rt.stream)
        return True

    def _sgd_pack_done(self):
        N, H, W = self._sgd_geom
        self._pairs_key = self._pack_key(H, W, N)

    def _pack_if_needed(self, H, W, N):
        key = self._pack_key(H, W, N)
        if key == self._packed_key:
            return False
        # ONE launch of pair jobs on the main stream; the single-form data-gradient matrices (segmented decoder levels) on the side
        # stream, joined at the start of backward.  (Packing the late 96 % of the parameters on the side stream beside the first
        # encoder levels was measured in round 3: 5.338 vs 5.356 ms, the pack is HBM-bound and so are those levels; removed.)
        main, late = self._tables(H, W, N)[1]
        main.run(skip_pairs=(key == self._pairs_key))       # (the optimizer step packed the pair jobs already: _sgd_pack)
        side = self.rt.side_stream() if late is not None else None
        self._dg_pack_on_side = side is not None
        if side is not None:
            nv.call('segnb_stream_fork', self.rt.stream, side.cuda_stream)
            with torch.cuda.stream(side):
                late.run()
            self._packed_key = key
            return True                              # (the side stream waits for everything issued before this forward)
        elif late is not None:
            late.run()
        self._packed_key = key
        return False

    _dg_pack_on_side = False

    def _dropout_tables(self, b, N, train):
        """Per-block [N, Cp] multiplier tables (0 or 1/(1-p)); None when Dropout2d is inactive.  All eleven
        tables live in one buffer drawn by ONE bernoulli launch per step (lib/models/zf_unet.py:25,31 draws
        one Dropout2d mask per block)."""
        names = ENCODER + DECODER
        ov = self.module.dropout_override
        if not train or (self.p_drop <= 0.0 and ov is None):
            return {n: None for n in names}
        tabs = b['drop']                         # {name: fp32 [N, Cp] view of one flat buffer}
        if ov is None:
            b['drop_flat'].bernoulli_(1.0 - self.p_drop).mul_(1.0 / (1.0 - self.p_drop))
            return tabs
        out = {}
        for n in names:
            src = ov.get(n)
            if src is None:
                out[n] = None
                continue
            t = tabs[n]
            t.fill_(1.0)
            t[:, :src.shape[1]] = src.to(self.rt.device, torch.float32)
            out[n] = t
        return out

    # ---- launch plans (segnb_plan_*): the forward / backward launch lists replayed from C ----------------------------
    # The first step of a configuration runs eagerly WHILE the library records the ABI calls; later steps replay the list
    # with one call (the Python launcher needs 10-14 us per launch: 3.6-4.4 ms for the ~300 launches of a step).  Valid
    # only while every pointer in the list is: same buffers (geometry, flat parameter / gradient storage), same streams,
    # same mode.  Anything a replay cannot express -- the data-parallel hooks between gradient groups, the bench's
    # per-launch timer, a uint8 input (host normalisation constants), CPU -- runs the eager path.
    use_cplan = os.environ.get('SEGNB_CPLAN', '1') != '0'

    def _cplan_key(self, kind, N, H, W, train, need_grad, drop):
        from segnb import engine
        rt = self.rt
        if not self.use_cplan or rt.device.type != 'cuda':
            return None
        side = rt.side_stream()
        return (kind, N, H, W, bool(train), bool(need_grad), tuple(n for n in ENCODER + DECODER if drop[n] is not None),
                getattr(self.module, '_grad_ready_hook', None) is not None,
                id(engine.TIMER),                     # (its event records are part of the list recorded under it)
                rt.stream, side.cuda_stream if side is not None else 0, self.flat.flat_p.data_ptr(),
                self.flat.flat_g.data_ptr(), tuple(p.data_ptr() for p in self.flat.buffer_list()))

    # ---- recorded launch lists (segnb_plan_*) ---------------------------------------------------------------------
    # A list is a sequence of segments [(handle, launches, mark)]: the data-parallel "gradients ready" hook is host code that
    # must run BETWEEN launches (it starts an all-reduce behind what has been issued so far), so the recording is cut there
    # and the replay calls the hook after the segment that ends at the cut.
    _rec = None

    def _plan_begin(self):
        from segnb import engine
        if engine.TIMER is not None:
            engine.TIMER.persistent = True
        self._rec = []
        nv.plan_record_begin()

    def _plan_cut(self, mark, resume=True):
        if self._rec is not None:
            handle, nops = nv.plan_record_end()
            self._rec.append((handle, nops, mark))
            self._paused = not resume
            if resume:
                nv.plan_record_begin()

    _paused = False

    def _plan_resume(self):
        if self._rec is not None and self._paused:
            self._paused = False
            nv.plan_record_begin()

    def _plan_end(self, ckey):
        handle, nops = nv.plan_record_end()
        segs, self._rec = self._rec + [(handle, nops, None)], None
        if any(h is None for h, _, _ in segs):           # a call that cannot be replayed: remembered, never recorded again
            for h, _, _ in segs:
                if h is not None:
                    nv.call('segnb_plan_destroy', h)
            segs = None
        self._cplans[ckey] = (segs, self._stage_state(), sum(n for _, n, _ in segs) if segs else 0)

    def _plan_abort(self, ckey):
        """An exception escaped a step that was being recorded (e.g. an allocation failure the caller catches): close the
        recording, free its segments and remember the configuration as eager."""
        if self._rec is None:
            return
        segs, self._rec = self._rec, None
        if not self._paused:
            nv.plan_record_abort()
        self._paused = False
        for h, _, _ in segs:
            if h is not None:
                try:
                    nv.call('segnb_plan_destroy', h)
                except Exception:
                    pass
        self._cplans[ckey] = (None, self._stage_state(), 0)

    def _plan_replay(self, plan, H=None, W=None):
        for handle, _, mark in plan[0]:
            nv.call('segnb_plan_run', handle)
            if mark is not None:
                hook = getattr(self.module, '_grad_ready_hook', None)
                if hook is not None:
                    los = self._tables(H, W)[3]
                    hook(self.flat, los[mark[1]], (self.rt.side_stream(),) if mark[2] else ())
        self._restore_stage_state(plan[1])

    def __del__(self):
        try:                       # recorded lists are owned by this plan: free them with it
            for plan in self._cplans.values():
                for handle, _, _ in (plan[0] or ()):
                    if handle is not None:
                        nv.call('segnb_plan_destroy', handle)
        except Exception:          # (interpreter shutdown: the library may be gone already)
            pass

    def _stage_state(self):
        return [(st, st._stats_stale, st._fused_fwd, getattr(st, '_saved', None))
                for n in ENCODER + DECODER for st in self.stages[n]]

    @staticmethod
    def _restore_stage_state(state):
        for st, stale, fused, saved in state:
            st._stats_stale, st._fused_fwd, st._saved = stale, fused, saved

    # ---- forward ---------------------------------------------------------------------------------------
    def forward(self, x, train, need_grad):
        # (SEGNB_REPLAY_GUARD=1: a replayed forward must execute the launches of the forward that recorded its lists)
        # The census starts where the two paths part (the list look-up): what precedes it -- Dropout2d tables, the weight pack when
        # the parameters changed, the gradient-buffer clear, the input pack -- is the same host code either way
        self._guard_mode, self._guard_active = (None, None), False
        out = self._forward(x, train, need_grad)
        self._guard_f.end(self._guard_active, *self._guard_mode)
        return out

    def _forward(self, x, train, need_grad):
        rt = self.rt
        self.flat.ensure(rt.device)
        u8 = x.dtype == torch.uint8
        if u8:
            N, H, W, C = x.shape
        else:
            N, C, H, W = x.shape
        b = self.buffers(N, H, W)
        drop = self._dropout_tables(b, N, train)      # (first: on the side stream, beside the weight pack below)
        forked = self._pack_if_needed(H, W, N)
        if train and need_grad:
            self.flat.prezero(rt, forked=bool(forked))
            # (the optimizer step that follows this step's backward may fuse its update with the next forward's pack)
            self._sgd_geom = (N, H, W)
            self.flat.sgd_pack_hook, self.flat.sgd_pack_done = self._sgd_pack, self._sgd_pack_done
        hf = self._head_fusable(train, need_grad)
        self._last_head_fused = hf
        ckey = None if u8 else self._cplan_key('fwd', N, H, W, train, need_grad, drop)
        first = None
        if u8 and self.stages[ENCODER[0]][0].conv.u8_direct_ok(N, H, W, self.wp[0]):
            first = (x, self.module.input_norm)          # the first convolution reads the image itself
        else:
            # the batch enters through ONE launch outside the recorded list (it reads the caller's tensor, whatever its
            # address): NCHW fp32 / NHWC uint8 -> the padded NHWC buffer the list's first convolution reads
            pack_input(rt, x, b['x'], self.module.input_norm)
        if ckey is not None:
            # statistics a fused training forward left unconsumed are cleared here, outside the recorded list
            for n in ENCODER + DECODER:
                for st in self.stages[n]:
                    if st.bn is not None and train and st._stats_stale:
                        st.stats.zero_()
                        st._stats_stale = False
            plan = self._cplans.get(ckey)
            self._guard_active = self._guard_f.begin()
            if plan is not None and plan[0] is not None:
                self._guard_mode = (ckey, 'replay')
                self._plan_replay(plan)
                self._last = (N, H, W) if need_grad else None
                self._last_train = bool(train)
                self.generation += 1
                return self._head(b, N, H, W, hf)
            if plan is None:
                self._plan_begin()
            else:
                ckey = None                               # recorded before and found not replayable: eager
        try:
            wp = self.wp
            cur = b['x']
            for i, name in enumerate(ENCODER):
                s1, s2 = self.stages[name]
                # consumer-side BatchNorm: where the block's second convolution can apply the first one's BatchNorm + ReLU
                # while it loads (segnb_conv_fprop_tf: the thin 32 -> 32 level), that activated tensor is never written
                defer = self._defer(s1, s2, cur, b['a1_%d' % i], train, need_grad)
                y1 = s1.forward(cur, train, None, out=b['a1_%d' % i], need_grad=need_grad, u8=first if i == 0 else None,
                                defer_act=defer)
                x2, tf2 = (y1, s1.tf_out()) if defer else (b['a1_%d' % i], None)
                if i < 5:
                    skip = b['cat_%d' % i].slice(wp[i + 1], wp[i])
                    s2.forward(x2, train, drop[name], out=skip, pool_out=b['p_%d' % (i + 1)], need_grad=need_grad, x_tf=tf2)
                    cur = b['p_%d' % (i + 1)]
                else:
                    s2.forward(b['a1_%d' % i], train, drop[name], need_grad=need_grad,
                               up_out=b['cat_4'].slice(0, wp[5]) if self._upsampled(4, N, H, W) else None, out=b.get('u_4'))
            for name, lvl in zip(DECODER, (4, 3, 2, 1, 0)):
                s1, s2 = self.stages[name]
                if self.subpixel:
                    s1.conv.bind_up(b.get('u_%d' % lvl), b.get('du_%d' % lvl))
                defer = self._defer(s1, s2, b['cat_%d' % lvl], b['b1_%d' % lvl], train, need_grad)
                y1 = s1.forward(b['cat_%d' % lvl], train, None, out=b['b1_%d' % lvl], need_grad=need_grad, defer_act=defer)
                x2, tf2 = (y1, s1.tf_out()) if defer else (b['b1_%d' % lvl], None)
                if lvl > 0:
                    s2.forward(x2, train, drop[name], need_grad=need_grad,
                               up_out=(b['cat_%d' % (lvl - 1)].slice(0, wp[lvl]) if self._upsampled(lvl - 1, N, H, W) else None),
                               out=b.get('u_%d' % (lvl - 1)), x_tf=tf2)
                elif hf:
                    # the last activation pass runs with the classifier (_head): the activated tensor is never written
                    s2.forward(x2, train, drop[name], need_grad=need_grad, x_tf=tf2, defer_act='head')
                else:
                    s2.forward(x2, train, drop[name], out=b['f0'], need_grad=need_grad, x_tf=tf2)
        except BaseException:
            if ckey is not None:
                self._plan_abort(ckey)
            raise
        if ckey is not None:
            self._plan_end(ckey)
            if self._cplans[ckey][0] is not None:
                self._guard_mode = (ckey, 'record')
        self._last = (N, H, W) if need_grad else None
        self.generation += 1                       # every forward overwrites the activation buffers
        self._last_train = bool(train)
        return self._head(b, N, H, W, hf)

    # SEGNB_HEAD_FUSION=0: the last BatchNorm / activation pass and the classifier as separate launches (A/B)
    HEAD_FUSION = os.environ.get('SEGNB_HEAD_FUSION', '1') != '0'

    def _head_fusable(self, train, need_grad):
        """A differentiated training forward whose last stage folds its finalize: that stage's activation pass and the 1x1
        classifier are ONE launch (segnb_bn_fwd_fused_head), and so are the classifier's backward and the stage's
        BatchNorm-backward reduction (segnb_head_bn_bwd) -- the activated tensor and its gradient never exist in memory."""
        s2 = self.stages[DECODER[-1]][1]
        return bool(self.HEAD_FUSION and train and need_grad and s2.bn is not None and s2.fuse_finalize
                    and nv.query('segnb_head_fused_ok', self.K, self.wp[0]))

    def _defer(self, s1, s2, x1, a1, train, need_grad):
        """May the block's first stage skip its activation pass (its second convolution and that one's weight gradient
        apply BatchNorm + ReLU to the raw output while they load)?"""
        if self.rt.code != nv.BF16 or not s1.defer_act_ok(x1, train, need_grad) or not isinstance(s2.conv, ConvOp):
            return False
        y = s1.buffers(x1.N, a1.H, a1.W)['y']
        y2 = s2.buffers(x1.N, a1.H, a1.W)['y']
        return s2.conv.fprop_tf_ok(y, y2) and s2.conv.wgrad_tf_ok(y, y2)

    def _head(self, b, N, H, W, fused=False):
        """The 1x1 classifier (zf_unet.py:58) as ONE launch outside the recorded list, straight into the tensor the caller
        gets (the list would have to write a persistent buffer and the caller's copy would be a launch of its own); the
        loss is told where this model wants d(loss)/d(logits) (segnb.seglosses.register_grad_buffer)."""
        from segnb import seglosses
        rt = self.rt
        head = self.module.conv_final
        logits = torch.empty((N, self.K, H, W), dtype=torch.float32, device=rt.device)
        if fused:
            self.stages[DECODER[-1]][1].head_forward(head.weight.detach(), head.bias.detach(), self.K, logits)
        else:
            nv.call('segnb_head_fwd', rt.code, b['f0'].ptr, b['f0'].ld, N, H, W, self.widths[0],
                    nv.ptr(head.weight.detach()), nv.ptr(head.bias.detach()), self.K, nv.ptr(logits), rt.stream)
        seglosses.register_grad_buffer(logits, b['dlogits_in'])
        return logits

    # ---- backward --------------------------------------------------------------------------------------
    def backward(self, dlogits):
        self._guard_mode, self._guard_active = (None, None), False
        from segnb import engine as _engine
        try:
            out = self._backward(dlogits)
        finally:
            _engine.DW_OVERWRITE = False
        self._guard_b.end(self._guard_active, *self._guard_mode)
        return out

    def _backward(self, dlogits):
        rt, flat, wp = self.rt, self.flat, self.wp
        if self._last is None:
            raise RuntimeError('backward without a grad-enabled forward')
        if not self._last_train and self.module._cfg['bn']:
            # the backward plan implements the TRAINING-mode BatchNorm gradient (batch statistics); through an eval-mode
            # forward (running statistics) BatchNorm is a plain affine map and that formula is wrong
            raise RuntimeError('backward through an eval-mode forward is not supported: BatchNorm gradients are '
                               'implemented for training mode (call model.train(), or run the forward under no_grad)')
        N, H, W = self._last
        b = self.buffers(N, H, W)
        accumulate_in_place = flat.begin_backward()
        from segnb import engine as _engine
        _engine.DW_OVERWRITE = not accumulate_in_place           # (reset by backward(): see ConvOp._arm_target)
        drop_now = {n: self.stages[n][1]._saved[2] if self.stages[n][1]._saved is not None else None for n in ENCODER + DECODER}
        ckey = self._cplan_key('bwd', N, H, W, True, True, drop_now)
        if ckey is not None:
            ckey = ckey + (bool(getattr(self, '_last_head_fused', False)), bool(accumulate_in_place))
        if ckey is not None:
            din = b['dlogits_in']
            if dlogits.data_ptr() != din.data_ptr():
                din.copy_(dlogits)                        # (a loss that did not write into the registered buffer)
            dlogits = din
            plan = self._cplans.get(ckey)
            self._guard_active = self._guard_b.begin()
            if plan is not None and plan[0] is not None:
                self._guard_mode = (ckey, 'replay')
                self._plan_replay(plan, H, W)
                rt._side_busy = False
                self._after_backward()
                flat.publish_grads(accumulate_in_place)
                return [None] * len(flat._off)
            if plan is None:
                self._plan_begin()
            else:
                ckey = None
        try:
            if self._dg_pack_on_side and rt.side_stream() is not None:
                # the data gradients' matrices were packed on the side stream (recorded: a replayed list waits too)
                nv.call('segnb_stream_join', rt.stream, rt.side_stream().cuda_stream)
            head = self.module.conv_final
            hf = bool(getattr(self, '_last_head_fused', False))
            if hf:
                self.stages[DECODER[-1]][1].head_backward(head.weight.detach(), self.K, dlogits, flat.grad_of(head.weight),
                                                          flat.grad_of(head.bias))
            else:
                nv.call('segnb_head_bwd', rt.code, b['f0'].ptr, b['f0'].ld, N, H, W, self.widths[0], wp[0],
                        nv.ptr(head.weight.detach()), self.K, nv.ptr(dlogits), b['df0'].ptr, b['df0'].ld,
                        nv.ptr(flat.grad_of(head.weight)), nv.ptr(flat.grad_of(head.bias)), rt.stream)
            # (Holding the weight gradients of the first decoder levels back until the dependent chain has reached the deep levels
            # -- HBM-bound launches beside MFMA-bound ones -- was measured in rounds 1 and 3: the chain's convolutions then stretch
            # by 6 %, step +1 %; so was holding the last block's second weight gradient back at the tail: +-0.  Both removed.)
            for name, lvl in zip(reversed(DECODER), (0, 1, 2, 3, 4)):
                s1, s2 = self.stages[name]
                # the first convolution of a block has ONE direct gradient source -- the data gradient of the second one: that
                # launch also does its BatchNorm-backward reduction where a fused kernel serves the shape (fuse_reduce_of)
                if lvl == 0 and hf:
                    red = s2.backward(flat, dx=b['db1_0'], fuse_reduce_of=s1, dz_ready=True)
                elif lvl == 0:
                    red = s2.backward(flat, g_direct=b['df0'], dx=b['db1_0'], fuse_reduce_of=s1)
                elif self._seg(lvl - 1, N, H, W):
                    # the level above handed the gradient of this block's output over at THIS resolution (du)
                    red = s2.backward(flat, g_direct=b['du_%d' % (lvl - 1)], dx=b['db1_%d' % lvl], fuse_reduce_of=s1)
                else:
                    red = s2.backward(flat, g_up=b['dcat_%d' % (lvl - 1)].slice(0, wp[lvl]), dx=b['db1_%d' % lvl],
                                      fuse_reduce_of=s1)
                if self.subpixel:
                    s1.conv.bind_up(b.get('u_%d' % lvl), b.get('du_%d' % lvl))
                s1.backward(flat, g_direct=b['db1_%d' % lvl], dx=b['dcat_%d' % lvl], reduced=red)
            self._unpack_group(H, W, 0)           # decoder (+ the head's gradients, written above on this stream)
            for i in (5, 4, 3, 2, 1, 0):
                if i == 3:
                    self._unpack_group(H, W, 1)   # conv_7 / conv_14: the bulk of the encoder's parameters
                s1, s2 = self.stages[ENCODER[i]]
                if i == 5 and self._seg(4, N, H, W):
                    red = s2.backward(flat, g_direct=b['du_4'], dx=b['da1_5'], fuse_reduce_of=s1)
                elif i == 5:
                    red = s2.backward(flat, g_up=b['dcat_4'].slice(0, wp[5]), dx=b['da1_5'], fuse_reduce_of=s1)
                else:
                    red = s2.backward(flat, g_direct=b['dcat_%d' % i].slice(wp[i + 1], wp[i]), g_pool=b['dp_%d' % (i + 1)],
                                      dx=b['da1_%d' % i], fuse_reduce_of=s1)
                s1.backward(flat, g_direct=b['da1_%d' % i], dx=(b['dp_%d' % i] if i > 0 else None), reduced=red)
            rt.join_side()                        # the weight gradients ran on the side stream
            self._tables(H, W)[2][2].run()       # the remaining packed weight-gradient workspaces -> flat gradient buffer
        except BaseException:
            if ckey is not None:
                self._plan_abort(ckey)
            raise
        if ckey is not None:
            self._plan_end(ckey)
            if self._cplans[ckey][0] is not None:
                self._guard_mode = (ckey, 'record')
        self._after_backward()
        # gradients live in ONE flat buffer; parameter.grad tensors are views of it (installed here, not
        # returned through autograd, so they never get cloned and a flat optimizer / all-reduce can run)
        flat.publish_grads(accumulate_in_place)
        return [None] * len(flat._off)

    def _after_backward(self):
        hook = getattr(self.module, '_grad_sync_hook', None)
        if hook is not None:
            hook(self.flat)
