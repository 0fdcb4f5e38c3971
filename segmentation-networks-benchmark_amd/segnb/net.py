"""Define-by-run executor over the C-ABI primitives, with automatic backward.

A model's ``_build(tape, x)`` is ordinary imperative code that calls the layer helpers below (conv_unit,
bn_act, maxpool, add, concat, head_1x1 ...).  Every helper launches its forward kernels immediately and
records a backward closure on the tape; ``Tape.backward`` runs the closures in reverse.  Buffers, packed
weights and tap tables are cached per call site (the graphs are static), so steady-state steps allocate nothing.

Gradients of multi-consumer tensors (ResNet identity branches, dense concatenations, skip adds) are combined by
``contribute``: the first consumer's gradient buffer is adopted as is, later ones are added in place
(segnb_add).  torch.cat is zero-copy as in the ZF_UNET plan: producers write into channel slices of one buffer.

Used by lib.models.{unet16, linknet, tiramisu}; ZF_UNET keeps its hand-scheduled plan (fused pool/upsample
routing, batched pack/unpack).
"""
import os

import numpy as np
import torch
from torch import nn

from . import _native as nv
from . import convplan as cp
from .engine import (BN_EPS, BN_MOMENTUM, STAT_REPLICAS, ConvOp, FlatParams, InputNorm, PackTable, Runtime, View,
                     pack_input, vld, vptr)


class Act(object):
    """Activation tensor: forward view + gradient view accumulated during backward.

    consumers: how many operators read this tensor in the forward (Tape.consume); producer: set by conv_unit when the tensor is
    act(BatchNorm(conv)) with nothing else in the way -- (y View, coef, sums, act, slope), what segnb_conv_fprop_bnreduce needs to
    do that layer's BatchNorm-backward reduction in the epilogue of the ONE consumer's data gradient; g_is_dz: that happened
    (the producer's sums are complete; .g is still the plain gradient, the direct apply form recomputes dz from it).
    coef None: the tensor is act(conv) WITHOUT BatchNorm (y = the activated tensor itself); a consumer that can apply act' while it
    writes the gradient (the classifier heads: segnb_head_conv_bwd) then stores dz itself and sums it into `sums` -- g_is_dz then
    means .g IS dz and the producer's segnb_bn_act_bwd_reduce pass is skipped."""
    __slots__ = ('v', '_g', '_g2', '_settle', 'needs_grad', 'consumers', 'producer', 'g_is_dz', 'lazy_ok', 'lazy_dgrad')

    def __init__(self, v, needs_grad=True):
        self.v, self._g, self._g2, self._settle, self.needs_grad = v, None, None, None, needs_grad
        self.consumers, self.producer, self.g_is_dz = 0, None, False
        # lazy_ok: the producer (a pre-activation BatchNorm, _bn_act_core) can take its gradient as "recompute it": lazy_dgrad =
        # (ConvOp, dy View) set by the ONE consumer's _data_gradient, which then only ran the sums launch -- the gradient tensor
        # itself never exists (segnb_conv_fprop_bnsums / _bnapply)
        self.lazy_ok, self.lazy_dgrad = False, None

    # The gradient.  A SECOND contribution is held back (Tape.contribute: _g2 + the launch that would add it): the convolution units
    # hand both sources to their reduction pass (segnb_bn_act_bwd_reduce_add) instead of an add pass over three tensors; any other
    # reader of .g gets the sum -- reading it launches the add.
    @property
    def g(self):
        if self._g2 is not None:
            settle, self._g2, self._settle = self._settle, None, None
            settle()
        return self._g

    @g.setter
    def g(self, v):
        assert self._g2 is None
        self._g = v

    def take_sources(self):
        """-> (g, g2 or None) WITHOUT adding them; the caller consumes both."""
        g, g2 = self._g, self._g2
        self._g2 = self._settle = None
        return g, g2


class _HostStep(object):
    """What a cut (or the end) of a backward launches from the host, outside the recorded lists: the batched bias gradients of
    the convolutions without BatchNorm, then the batched unpack of the weight gradients.  run() repeats it in a replayed step.
    end_side: this is the END of a backward whose weight gradients ran on the side stream -- the unpack goes behind the last of
    them on that stream and the two streams join after it (Tape.finish)."""

    def __init__(self, rt, bias, unpack, end_side=False, join=False):
        self.rt, self.bias, self.unpack, self.end_side, self.join = rt, bias, unpack, end_side, join

    def run(self):
        if self.bias is not None:
            nv.call('segnb_bias_grad_multi', nv.ptr(self.bias[0]), self.bias[1], self.rt.stream)
        side = self.rt.side_stream() if self.join else None
        if side is not None and not self.end_side:
            nv.call('segnb_stream_join', self.rt.stream, side.cuda_stream)
        if self.unpack is not None:
            if self.end_side:
                with torch.cuda.stream(side):
                    self.unpack.run()
            else:
                self.unpack.run()
        if side is not None and self.end_side:
            nv.call('segnb_stream_join', self.rt.stream, side.cuda_stream)


class Tape(object):
    def __init__(self, module, device, dtype):
        self.module = module
        self.rt = Runtime(device, dtype)
        self.flat = FlatParams(module)
        self.flat.ensure(self.rt.device)
        self._cache = {}
        self.pack_key = None
        self.back = []
        # every (convolution, input size) the model has run, in first-use order: ONE batched weight-pack launch per
        # parameter update and ONE batched gradient-unpack launch per backward (PackTable), instead of two / one per layer
        self.convs = []
        self._conv_seen = set()
        self._pack_table = None
        self._unpack_tables = {}
        self._unpack_pending = []
        self._bias_pending, self._bias_tables = [], {}
        self._zero_each_step = {}      # key -> tensor cleared by begin() (statistics tables several layers read: nobody's backward owns them)
        self._cuts = {}
        self._drop_sites, self._drop_pools = {}, {}
        self.plans = {}            # recorded launch lists (HipNet._run / _run_backward)
        # forward statistics that segnb_bn_fwd_fused left for this step's backward to clear (segnb_bn_bwd_apply_fused*):
        # if that backward never runs, begin() clears them before the next forward accumulates on top
        self.fused_stats, self.stats_pending = [], False
        # share (%) of the CUs the model's weight gradients split their pixels for (HipNet.wg_cu_pct; None = the library default)
        self.wg_cu_pct = getattr(module, 'wg_cu_pct', None)
        if os.environ.get('SEGNB_WG_CU_PCT'):          # (A/B: overrides the model's share)
            self.wg_cu_pct = int(os.environ['SEGNB_WG_CU_PCT'])
        self.lazy_add = bool(getattr(module, 'lazy_add', False))

    # BatchNorm finalize folded into the activation / apply launches of a differentiated training forward (one launch less
    # per layer and direction; A/B: SEGNB_FUSE_FINALIZE=0)
    fuse_finalize = os.environ.get('SEGNB_FUSE_FINALIZE', '1') != '0'

    def fuses_finalize(self):
        return self.fuse_finalize and self.train and self.need_grad

    def note_fused_stats(self, stats):
        self.fused_stats.append(stats)

    def __del__(self):
        try:                       # recorded launch lists are owned by this tape: free them with it
            for ent in self.plans.values():
                for h in [ent.get('fwd')] + [h for h, _ in (ent.get('bwd') or [])]:
                    if h:
                        nv.call('segnb_plan_destroy', h)
        except Exception:          # (interpreter shutdown: the library may be gone already)
            pass

    # ---- per-step bookkeeping ------------------------------------------------------------------------------
    def begin(self, train, need_grad):
        self.flat.ensure(self.rt.device)
        if self.stats_pending:                      # the differentiated forward before this one had no backward
            for t in self.fused_stats:
                t.zero_()
            self.stats_pending = False
        self.train, self.need_grad = train, need_grad
        for t in self._zero_each_step.values():
            t.zero_()
        self.back, self._seq = [], 0
        self._adopted_ranges = []          # (lazy_add: gradient views taken as accumulating first contributions, this step)
        self._bias_pending = []
        self.fused_stats = []
        for p, pool in self._drop_pools.items():
            pool['drawn'] = pool['used'] if train else 0
            if train and pool['used']:
                pool['buf'][:pool['used']].bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
        self.generation = getattr(self, 'generation', 0) + 1
        # weight-packing generation: every ConvOp plan (one per input size) remembers the generation it was packed at,
        # so a plan first used -- or last used -- under other parameter values is (re)packed on its next use
        key = (sum(p._version for p in self.flat.param_list()), self.flat.version, self.flat.flat_p.data_ptr())
        if key != self.pack_key and self.convs:
            # parameters changed since the last pack: all known plans in one launch (plans first met later in this
            # forward pack themselves in conv_unit and join the table for the next step)
            t = self._pack_table
            if t is None or t[0] != (len(self.convs), self.flat.flat_p.data_ptr()):
                jobs = []
                for conv, h, w in self.convs:
                    jobs += conv.pack_jobs(h, w)
                t = self._pack_table = ((len(self.convs), self.flat.flat_p.data_ptr()),
                                        PackTable(self.rt, jobs, 'segnb_pack_weight_multi', 'segnb_pack_weight'))
            t[1].run()
            for conv, h, w in self.convs:
                conv.plan(h, w)['packed_key'] = key
        self.pack_key = key

    def site(self, tag):
        """Call-site identity = position in the (static) build order."""
        self._seq += 1
        return '%s#%d' % (tag, self._seq)

    def cached(self, key, make):
        v = self._cache.get(key)
        if v is None:
            v = make()
            self._cache[key] = v
        return v

    def view(self, key, N, H, W, Cp):
        return self.cached((key, N, H, W, Cp), lambda: View.alloc(self.rt, N, H, W, Cp))

    def small(self, key, shape, dtype):
        return self.cached((key, tuple(shape)), lambda: self.rt.zeros(shape, dtype))

    def step_zeroed(self, key, shape, dtype):
        """A small buffer that is zero at the start of every step (cleared in begin(), outside any recorded list)."""
        k = (key, tuple(shape))
        t = self._zero_each_step.get(k)
        if t is None:
            t = self._zero_each_step[k] = self.rt.zeros(shape, dtype)
        return t

    def record(self, fn):
        if self.need_grad:
            self.back.append(fn)

    # fuse_reduce = False (class attribute): every BatchNorm-backward reduction as a pass of its own (A/B)
    fuse_reduce = True
    # fold_head_mask = False: the classifier heads return the plain gradient and the layer before them masks it in its own pass (A/B)
    fold_head_mask = True
    # fuse_act_pool = False: conv -> ReLU -> MaxPool2d(2) without BatchNorm as convolution + one pass that activates and pools (A/B)
    fuse_act_pool = True
    inline_last_wgrad = os.environ.get('SEGNB_INLINE_LAST_WGRAD', '1') != '0'
    # two_launch_dgrad = True: a dense layer's 16 -> prefix data gradient is never stored -- one launch for the BatchNorm-backward sums,
    # one that recomputes it and applies the BatchNorm backward (segnb_conv_fprop_bnsums / _bnapply).  OFF: measured on MI355X
    # (profiles/r05_ab.txt) FCDenseNet103 16.23 ms per step with it against 15.74 without -- the general kernel is not a streaming
    # engine at K = 144 (its plain launch writes at 1.8 TB/s, the reduction store pass adds 70 % whether or not it stores), so running
    # it twice costs more than the two tensor transits it saves
    two_launch_dgrad = False

    def consume(self, *acts):
        """An operator of the forward reads these tensors (each will receive one gradient contribution from it)."""
        for a in acts:
            if a is not None:
                a.consumers += 1

    # lazy_add (per model: HipNet.lazy_add, default off): a SECOND gradient contribution to a tensor is held back for the reduction
    # pass of the layer that produced it instead of being added at once.  Only for models whose gradient views never alias
    # (FCDenseNet accumulates its dense blocks' contributions into slices of shared buffers, in place and in order: not there)
    lazy_add = False

    def _adopted(self, gview):
        """Does gview overlap a view some tensor took as its first gradient contribution (and accumulates into in place)?"""
        lo, hi = self._span(gview)
        for (p0, p1) in self._adopted_ranges:
            if lo < p1 and p0 < hi:
                return True
        return False

    @staticmethod
    def _span(v):
        return v.ptr, v.ptr + ((v.N * v.H * v.W - 1) * v.ld + v.Cp) * v.t.element_size()

    def contribute(self, act, gview):
        if not act.needs_grad:
            return
        if act._g is None:
            act.g = gview
            if self.lazy_add:
                self._adopted_ranges.append(self._span(gview))
            return

        def add(v=act._g):
            nv.call('segnb_add', self.rt.code, v.ptr, v.ld, gview.ptr, gview.ld, v.ptr, v.ld, v.N, v.H, v.W, v.Cp,
                    self.rt.stream)
        if self.lazy_add and act._g2 is None and not act.g_is_dz and not self._adopted(gview):
            # held back: gview is a buffer of its call site that nothing writes again during this backward -- which is NOT true of
            # a view another tensor adopted as its accumulating first contribution (add() / concat() hand the same gradient view,
            # or slices of it, to several inputs: ADVICE r5); such a view is added at once
            act._g2, act._settle = gview, add
        else:
            act.g                      # (a third contribution: the pending one is added first, in arrival order)
            add()

    def sources(self, act, two_ok):
        """-> (g, g2): the gradient of act as one or (two_ok, a held-back second contribution) two same-size views; without two_ok the
        pending add runs now."""
        if two_ok:
            return act.take_sources()
        return act.g, None

    def register_conv(self, conv, H, W):
        k = (id(conv), H, W)
        if k not in self._conv_seen:
            self._conv_seen.add(k)
            self.convs.append((conv, H, W))

    def defer_unpack(self, conv, H, W, grad_w):
        """conv.wgrad(..., unpack=False) was launched: its packed result joins this backward's batched unpack."""
        self._unpack_pending.append((conv, H, W, grad_w))

    def backward(self):
        self.run_closures(join=False)
        self.finish()

    # unpack_on_side = False (class attribute): the end-of-backward unpack runs on the main stream, after the join (A/B)
    unpack_on_side = True
    # early_unpack = False: no partial unpacks in the middle of a single-GPU backward (the cuts of CUT_FRACTIONS, below) (A/B)
    early_unpack = True

    def finish(self):
        """End of a backward whose closures ran with join=False: the batched bias gradients on the current stream; the batched
        unpack of the weight gradients BEHIND the last of them on the side stream (the dependent chain usually ends first: on the
        main stream the unpack -- 0.1-0.27 ms for LinkNet34 / UNet16 -- was a serial tail of the step), then the join.
        -> the object whose run() repeats this in a replayed step."""
        rt = self.rt
        busy = bool(getattr(rt, '_side_busy', False)) and rt.side_stream() is not None
        on_side = busy and self.unpack_on_side
        bias = self._run_bias_grads('end')
        if on_side:
            with torch.cuda.stream(rt.side_stream()):
                step = self._run_unpack_tables('end')
        else:
            rt.join_side()
            step = self._run_unpack_tables('end')
        rt.join_side()
        if bias is None and step is None and not busy:
            return None
        return _HostStep(rt, bias, step, end_side=on_side, join=busy)

    # ---- gradients handed to the data-parallel hook while backward is still running -----------------------------------
    # Parameters sit in the flat buffer in registration (= forward) order and the closures run in reverse, so the END of
    # the flat gradient buffer is final first.  The first backward under an active data-parallel hook learns, per closure,
    # the frontier "every gradient at flat offset >= lo is final after this closure" (FlatParams.touch_log: which
    # parameters a closure writes); later backwards stop at the closures where the frontier passes 2/3 and 1/3 of the
    # buffer, unpack the weight gradients launched so far (behind them, on the weight-gradient stream) and call
    # model._grad_ready_hook(flat, lo, producers) -- segnb.dist.DataParallel.grads_ready starts the all-reduce of the
    # finished buckets beside the rest of backward (ZF_UNET does the same with its fixed groups).
    CUT_FRACTIONS = (2.0 / 3.0, 1.0 / 3.0)

    def _ready_hook(self):
        hook = getattr(self.module, '_grad_ready_hook', None)
        if hook is None or not getattr(getattr(hook, '__self__', None), 'active', True):
            return None
        return hook

    def _learn_cuts(self, nback, frontiers):
        cuts, k = {}, 0
        for i, lo in enumerate(frontiers):
            while k < len(self.CUT_FRACTIONS) and lo <= self.CUT_FRACTIONS[k] * self.flat.total:
                if 0 < lo < self.flat.total:
                    cuts[i] = lo           # (several fractions passed by one closure: the lowest offset wins)
                k += 1
        # (one more cut in front of the LAST closure that writes gradients -- so that the tail behind the first layer's weight
        # gradient is the unpack of that layer alone -- measured +-0 on LinkNet34 and UNet16, same box: not kept)
        self._cuts[nback] = cuts

    def run_closures(self, around_cut=None, join=True):
        """around_cut(do): wraps the partial unpack + hook of a cut -- do() performs them and returns (unpack table or None,
        flat offset, weight-gradient stream in use) -- so that HipNet can cut its recorded launch list there.
        join=False: the caller joins the weight-gradient stream itself (Tape.finish)."""
        hook = self._ready_hook()
        back = list(reversed(self.back))
        nback = len(back)
        # (without a data-parallel hook the cuts still pay: the weight gradients finished so far are unpacked on the side stream in
        # the middle of backward instead of in one serial tail behind the last of them -- Tape.early_unpack)
        want = hook is not None or self.early_unpack
        cuts = self._cuts.get(nback) if want else None
        learn = want and cuts is None
        if learn:
            flat = self.flat
            touches = []
        for i, fn in enumerate(back):
            if learn:
                flat.touch_log = set()
            fn()
            if learn:
                touches.append(flat.touch_log)
                flat.touch_log = None
            elif cuts and i in cuts and i + 1 < nback:
                def do(i=i):
                    side = self.rt.side_stream() if getattr(self.rt, '_side_busy', False) else None
                    if side is not None:
                        with torch.cuda.stream(side):
                            table = self.run_unpack(group=i)
                    else:
                        table = self.run_unpack(group=i)
                    if hook is not None:
                        hook(self.flat, cuts[i], (side,) if side is not None else ())
                    return table, cuts[i], side is not None
                if around_cut is not None:
                    around_cut(do)
                else:
                    do()
        if learn:
            # frontier after closure i: every gradient at a flat offset >= it is final.  Parameters no closure writes (a convolution
            # bias in front of a BatchNorm: its gradient is the cleared buffer's zero) are final from the start
            offs = sorted((flat._off[k][0], k) for k in flat._off)
            written = set().union(*touches) if touches else set()
            pending = {k for _, k in offs if k in written}
            frontiers, hi = [], len(offs)
            for t in touches:
                pending -= t
                while hi > 0 and offs[hi - 1][1] not in pending:
                    hi -= 1
                frontiers.append(offs[hi][0] if hi < len(offs) else flat.total)
            self._learn_cuts(nback, frontiers)
        self.back = []
        self.stats_pending = False
        if join:
            self.rt.join_side()           # the weight gradients ran on the side stream

    # batch_bias = False (class attribute): one segnb_bn_bwd_finalize launch per bias gradient (A/B)
    batch_bias = True

    def defer_bias_grad(self, sums, C, Cp, gb, count, coef_buf, bcoef):
        """The bias gradient of a convolution without BatchNorm (sum of dz, accumulated in `sums` by the reduction pass): joined to
        the ONE segnb_bias_grad_multi launch that goes with the next batched unpack (nothing reads a bias gradient earlier)."""
        if not self.batch_bias:
            nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, count, None, nv.ptr(coef_buf), nv.ptr(bcoef), None, nv.ptr(gb), 1,
                    self.rt.stream)
            return
        self._bias_pending.append((sums, C, Cp, gb))

    def _run_bias_grads(self, group):
        """-> the (job table, job count) launched, or None.  The caller keeps it: a replayed backward launches the same table
        from its host-side step (HipNet._run_backward) -- the call is made outside the recorded lists, like the unpack."""
        if not self._bias_pending:
            return None
        key = tuple((s.data_ptr(), 0 if g is None else g.data_ptr()) for s, _, _, g in self._bias_pending)
        t = self._bias_tables.get(group)
        if t is None or t[0] != key:
            dt = np.dtype([('sums', '<u8'), ('gb', '<u8'), ('C', '<i4'), ('Cp', '<i4')])
            assert dt.itemsize == nv.query('segnb_bias_grad_job_bytes'), 'BiasGradJob layout drifted from the ABI'
            rows = np.zeros(len(self._bias_pending), dtype=dt)
            for i, (sm, C, Cp, gb) in enumerate(self._bias_pending):
                rows[i] = (sm.data_ptr(), 0 if gb is None else gb.data_ptr(), C, Cp)
            tab = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy()).to(self.rt.device)
            t = self._bias_tables[group] = (key, tab, len(rows), list(self._bias_pending))
        nv.call('segnb_bias_grad_multi', nv.ptr(t[1]), t[2], self.rt.stream)
        self._bias_pending = []
        return t[1], t[2]

    def run_unpack(self, group='end'):
        """Batched unpack of the weight gradients launched since the last one (group: which cut of the backward this is), behind
        the batched bias gradients of the same span.  -> an object whose run() repeats both launches (a replayed step), or None."""
        bias = self._run_bias_grads(group)
        step = self._run_unpack_tables(group)
        if bias is None:
            return step
        return _HostStep(self.rt, bias, step)

    def _run_unpack_tables(self, group):
        if self._unpack_pending:
            key = (tuple((id(c), h, w) for c, h, w, _ in self._unpack_pending), self.flat.flat_g.data_ptr())
            t = self._unpack_tables.get(group)
            if t is None or t[0] != key:
                jobs = []
                for conv, h, w, gw in self._unpack_pending:
                    jobs += conv.unpack_jobs(h, w, gw)
                t = self._unpack_tables[group] = (key, PackTable(self.rt, jobs, 'segnb_unpack_wgrad_multi', 'segnb_unpack_wgrad'))
            t[1].run()
            self._unpack_pending = []
            return t[1]
        return None

    DROP_POOL_FLOATS = 1 << 20

    def dropout_table(self, site, N, Cp, p):
        """fp32 [N, Cp] Dropout2d multipliers (0 or 1/(1-p)) drawn on the device; None when inactive.  The tables of
        all call sites with the same p are slices of ONE pool, redrawn by one bernoulli launch per step in begin()
        (FCDenseNet103 has 100+ Dropout2d sites); a site first met in this step draws its own slice."""
        if not self.train or p <= 0.0:
            return None
        key = (site + '/drop', N, Cp)
        ent = self._drop_sites.get(key)
        if ent is None:
            pool = self._drop_pools.get(p)
            if pool is None:
                pool = self._drop_pools[p] = {'buf': self.rt.zeros((self.DROP_POOL_FLOATS,), torch.float32), 'used': 0,
                                              'drawn': 0}
            n = N * Cp
            if pool['used'] + n > pool['buf'].numel():
                raise RuntimeError('dropout pool exhausted (%d floats): raise Tape.DROP_POOL_FLOATS' % pool['buf'].numel())
            t = pool['buf'][pool['used']:pool['used'] + n].view(N, Cp)
            pool['used'] += n
            ent = self._drop_sites[key] = (t, pool['used'])
        t, end = ent
        if end > self._drop_pools[p]['drawn']:
            t.bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
        return t


# ------------------------------------------------------------------------------------------------------------
# layer helpers
# ------------------------------------------------------------------------------------------------------------
def _bn_fields(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, getattr(bn, 'num_batches_tracked', None),
            float(getattr(bn, 'eps', BN_EPS)), float(getattr(bn, 'momentum', BN_MOMENTUM) or BN_MOMENTUM))


def _data_gradient(tape, conv, x, dy, site):
    """dx of a convolution handed to its input.  When the input is the activated BatchNorm output of ONE convolution and this
    is its only consumer (Act.producer / .consumers), the launch's epilogue also does that layer's BatchNorm-backward reduction
    (segnb_conv_fprop_bnreduce, as ZF_UNET's second convolutions do: linknet.py:41-62 via dilated_resnet's BasicBlock): the
    producer then skips its reduction pass."""
    xv = x.v
    pr = x.producer
    if (pr is not None and pr[1] is not None and x.lazy_ok and tape.fuse_reduce and Tape.two_launch_dgrad and x.consumers == 1
            and x.g is None and x.needs_grad and conv.dgrad_bnapply_ok(dy, xv.N, xv.H, xv.W, xv.ld)):
        conv.dgrad_bnsums(dy, xv.H, xv.W, pr)
        x.lazy_dgrad = (conv, dy)
        x.g_is_dz = True
        return
    dx = tape.view(site + '/dx', xv.N, xv.H, xv.W, xv.Cp)
    if pr is not None and pr[1] is None:
        # activation without BatchNorm (linknet.py:58-61): the data gradient stores dz = g * act'(a) and sums it where a fused kernel
        # serves the shape; the producer then skips its mask pass (Act.g_is_dz)
        if (tape.fuse_reduce and Tape.fold_head_mask and x.consumers == 1 and x.g is None and x.needs_grad
                and conv.dgrad_actmask_ok(dy, dx)):
            conv.dgrad(dy, dx, bn_reduce=pr)
            x.g_is_dz = True
            tape.contribute(x, dx)
            return
        pr = None
    if (pr is not None and tape.fuse_reduce and x.consumers == 1 and x.g is None and x.needs_grad
            and conv.dgrad_bnreduce_ok(dy, dx)):
        yv, coef, sums, act, slope = pr
        conv.dgrad(dy, dx, bn_reduce=(yv, coef, sums, act, slope))
        x.g_is_dz = True
    else:
        conv.dgrad(dy, dx)
    tape.contribute(x, dx)


def _sum_into(tape, v, out_stats):
    """statistics of View v into the channel range out_stats = (table, element offset, row stride) (segnb_bn_stats_ld)"""
    table, off, ld = out_stats
    nv.call('segnb_bn_stats_ld', tape.rt.code, v.ptr, v.ld, v.N, v.H, v.W, v.Cp, nv.ptr(table, off), ld, tape.rt.stream)


def conv_unit(tape, x, weight, bias, in_segments, stride=1, pad=1, transposed=False, bn=None, act=nv.ACT_RELU,
              slope=0.01, dropmul=None, out=None, pool=False, pool_out=None, res=None, out_hw=None, tag='conv',
              out_stats=None):
    """conv / conv-transpose -> [BatchNorm] -> (+res) -> activation -> [Dropout2d multipliers] [-> MaxPool2d(2)].

    out: optional View to write the activated output into (a slice of a concat buffer).
    out_stats: (fp64 table, element offset, row stride) -- the per-channel statistics of the FINAL output (the pooled one when
    pool is set) are accumulated into that channel range (the slice's share of its concat buffer's statistics table: bn_act's
    stats_src), by the pass that writes it where there is one.
    Returns Act, or (Act, pooled Act) when pool is set."""
    rt = tape.rt
    site = tape.site(tag)
    def make_op():
        op = ConvOp(rt, weight, bias, in_segments, stride, pad, transposed, need_dgrad=x.needs_grad, out_hw=out_hw)
        op.wg_cu_pct = tape.wg_cu_pct          # (the model's share of the CUs for its weight gradients; None = the library default)
        return op
    conv = tape.cached(site + '/op', make_op)
    tape.consume(x, res)
    xv = x.v
    # The weight gradient of a layer WITHOUT a data gradient (the network's first convolution: nothing is left on the dependent chain
    # behind it) runs on the chain's own stream -- which is idle by then while the weight-gradient stream still works through its
    # queue (UNet16: 1.4 ms of it) -- instead of at the end of that queue (Tape.inline_last_wgrad = False: forked like the others, A/B).
    # Only where the launch delivers into the parameter's gradient itself (ConvOp.direct_ok): a result left in the packed workspace is
    # picked up by the batched unpack BEHIND the weight gradients on the side stream (Tape.finish), which does not wait for this one
    forks = x.needs_grad or not Tape.inline_last_wgrad or not conv.direct_ok()
    plan = conv.plan(xv.H, xv.W)
    tape.register_conv(conv, xv.H, xv.W)
    if plan.get('packed_key') != tape.pack_key:          # (first use of this plan: Tape.begin packs the known ones in one launch)
        conv.pack(xv.H, xv.W)
        plan['packed_key'] = tape.pack_key
    Ho, Wo = conv.out_hw(xv.H, xv.W)
    N, Cp, C = xv.N, conv.Cop, conv.Co
    y = tape.view(site + '/y', N, Ho, Wo, Cp)
    stats = tape.small(site + '/stats', (STAT_REPLICAS, 2, Cp), torch.float64)
    coef_buf = tape.small(site + '/coef', (4, Cp), torch.float32)
    has_bn = bn is not None
    use_batch_stats = has_bn and tape.train
    # InPlaceABN with the backend's affine form (lib/modules/abn: affine_form='abs_eps'): the kernels take |w| + eps as gamma
    # and hand dgamma to a scratch vector that segnb_abn_dscale signs into the parameter's gradient
    abs_form = has_bn and getattr(bn, 'affine_form', 'gamma') == 'abs_eps' and bn.weight is not None

    def eff_gamma():
        if not abs_form:
            return bn.weight.detach()
        geff = tape.small(site + '/geff', (C,), torch.float32)
        nv.call('segnb_abn_scale', nv.ptr(bn.weight.detach()), float(bn.eps), nv.ptr(geff), C, rt.stream)
        return geff

    def dgamma_target():
        return tape.small(site + '/dgeff', (C,), torch.float32) if abs_form else tape.flat.grad_of(bn.weight)

    def dgamma_done():
        if abs_form:
            nv.call('segnb_abn_dscale', nv.ptr(bn.weight.detach()), nv.ptr(tape.small(site + '/dgeff', (C,), torch.float32)),
                    nv.ptr(tape.flat.grad_of(bn.weight)), C, rt.stream)
    # activation in the convolution's epilogue (segnb_conv_fprop_act): no BatchNorm (unet16.py:12-21, the linknet head), or
    # BatchNorm in inference -- with no residual, dropout or fused pooling in the way.  The backward of the no-BatchNorm
    # form reads the ACTIVATED tensor where it read the raw one: act'(z) has the sign of act(z).
    ov_direct = out if out is not None else None
    # (a fused MaxPool2d(2) stays possible without BatchNorm under ReLU: the pooled tensor is then one more pass over the ACTIVATED
    # output -- read a, write a / 4 -- instead of the activation pass -- read y, write a and a / 4; unet16.py:113-118)
    fuse = ((not has_bn or (not tape.train and not tape.need_grad)) and res is None and dropmul is None
            and (not pool or (not has_bn and act == nv.ACT_RELU and Tape.fuse_act_pool))
            and act in (nv.ACT_NONE, nv.ACT_RELU, nv.ACT_LEAKY)
            and conv.act_epilogue_ok(xv.H, xv.W, N, out.ld if out is not None else Cp, bn))
    if fuse:
        coef = None
        if has_bn:
            gamma, beta, rm, rv, nbt, eps, mom = _bn_fields(bn)
            nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * Ho * Wo), nv.ptr(eff_gamma()),
                    nv.ptr(beta.detach()), eps, mom, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), 0, nv.ptr(coef_buf), rt.stream)
            coef = coef_buf
        ov = ov_direct if ov_direct is not None else tape.view(site + '/a', N, Ho, Wo, Cp)
        conv.fprop(xv, ov, None, epilogue=(coef, act, slope))
        pv = None
        if pool:
            pv = pool_out if pool_out is not None else tape.view(site + '/p', N, Ho // 2, Wo // 2, Cp)
            nv.call('segnb_maxpool_fwd', rt.code, ov.ptr, ov.ld, N, Ho, Wo, Cp, 2, 2, 0, pv.ptr, pv.ld, None, rt.stream)
        if out_stats is not None:
            _sum_into(tape, pv if pool else ov, out_stats)
        oa = Act(ov)
        pa = Act(pv) if pool else None
        if not has_bn and tape.need_grad and not pool:
            # (sums: zero between steps -- segnb_bias_grad_multi clears what it reads)
            oa.producer = (ov, None, tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64), act, slope)

        def backward_fused():
            gp = pa.g if pa is not None else None
            if oa._g is None and gp is None:
                return
            flat = tape.flat
            sums = tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64)
            bcoef = tape.small(site + '/bcoef', (3, Cp), torch.float32)
            if oa.g_is_dz:
                # the one consumer wrote dz = g * act'(a) and summed it (head_conv: segnb_head_conv_bwd)
                oa.g_is_dz = False
                dz = oa.g
            else:
                dz = tape.view(site + '/dz', N, Ho, Wo, Cp)
                g1, g2 = tape.sources(oa, gp is None)
                # dz = g * act'(a): the reduce pass with the activated tensor in the place of the raw one (the last launch before
                # the weight gradient's fork: its event rides on this dispatch, engine.Runtime.arm_fork); the pooled gradient is
                # routed to the window's first maximum of a, as the forward's pooling pass chose it
                (rt.arm_fork() if forks else None)
                if g2 is not None:
                    nv.call('segnb_bn_act_bwd_reduce_add', rt.code, ov.ptr, ov.ld, N, Ho, Wo, Cp, None, act, slope, None,
                            g1.ptr, g1.ld, g2.ptr, g2.ld, dz.ptr, dz.ld, nv.ptr(sums), None, 0, rt.stream)
                else:
                    nv.call('segnb_bn_act_bwd_reduce', rt.code, ov.ptr, ov.ld, N, Ho, Wo, Cp, None, act, slope, None,
                            vptr(g1), vld(g1), vptr(gp), vld(gp), None, 0, dz.ptr, dz.ld, nv.ptr(sums), None, 0, rt.stream)
            gb = flat.grad_of(bias) if bias is not None else None
            tape.defer_bias_grad(sums, C, Cp, gb, float(N * Ho * Wo), coef_buf, bcoef)
            side = rt.fork_side() if forks else None
            if side is not None:
                with torch.cuda.stream(side):
                    conv.wgrad(xv, dz, flat.grad_of(weight), unpack=False)
            else:
                conv.wgrad(xv, dz, flat.grad_of(weight), unpack=False)
            tape.defer_unpack(conv, xv.H, xv.W, flat.grad_of(weight))
            if x.needs_grad:
                _data_gradient(tape, conv, x, dz, site)

        if not has_bn:
            tape.record(backward_fused)
        return (oa, pa) if pool else oa
    # conv -> Dropout2d -> slice statistics (a dense layer of tiramisu.py:9-20) in ONE launch where segnb_conv_fprop_drop serves the
    # shape: the raw convolution output is never stored (its backward only needs the multipliers)
    drop_fused = (not has_bn and act == nv.ACT_NONE and dropmul is not None and not pool and res is None
                  and conv.drop_epilogue_ok(N, xv.H, xv.W, out.ld if out is not None else Cp))
    if drop_fused:
        ov = out if out is not None else tape.view(site + '/a', N, Ho, Wo, Cp)
        conv.fprop_drop(xv, ov, dropmul, out_stats)
        out_stats = None
        y = ov                  # (the reduction pass of the backward takes y for act'(z) only: ACT_NONE never reads it)
        coef, pv, fused_bn = None, None, False
    else:
        conv.fprop(xv, y, stats if use_batch_stats else None)
        coef = None
        ov = out if out is not None else tape.view(site + '/a', N, Ho, Wo, Cp)
        pv = (pool_out if pool_out is not None else tape.view(site + '/p', N, Ho // 2, Wo // 2, Cp)) if pool else None
        fused_bn = has_bn and tape.fuses_finalize()
        if fused_bn:
            # finalize + activation pass in one launch; the statistics stay for this layer's backward to clear, the
            # backward sums are cleared here (segnb_bn_fwd_fused / segnb_bn_bwd_apply_fused, include/segnb_hip.h)
            gamma, beta, rm, rv, nbt, eps, mom = _bn_fields(bn)
            sums_f = tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64)
            nv.call('segnb_bn_fwd_fused', rt.code, y.ptr, y.ld, N, Ho, Wo, C, Cp, nv.ptr(stats), nv.ptr(eff_gamma()),
                    nv.ptr(beta.detach()), eps, mom, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), nv.ptr(coef_buf), nv.ptr(sums_f),
                    act, slope, nv.ptr(dropmul), ov.ptr, ov.ld, vptr(pv), vld(pv), None, 0,
                    None if res is None else res.v.ptr, 0 if res is None else res.v.ld, rt.stream)
            tape.note_fused_stats(stats)
            coef = coef_buf
        else:
            if has_bn:
                gamma, beta, rm, rv, nbt, eps, mom = _bn_fields(bn)
                nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * Ho * Wo), nv.ptr(eff_gamma()),
                        nv.ptr(beta.detach()), eps, mom, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), 1 if tape.train else 0,
                        nv.ptr(coef_buf), rt.stream)
                coef = coef_buf
            if out_stats is not None and pv is None and res is None:
                # the pass that writes the slice also sums it
                nv.call('segnb_bn_act_fwd_stats', rt.code, y.ptr, y.ld, N, Ho, Wo, Cp, nv.ptr(coef), act, slope, nv.ptr(dropmul),
                        ov.ptr, ov.ld, nv.ptr(out_stats[0], out_stats[1]), out_stats[2], rt.stream)
                out_stats = None
            else:
                nv.call('segnb_bn_act_fwd', rt.code, y.ptr, y.ld, N, Ho, Wo, Cp, nv.ptr(coef), act, slope, nv.ptr(dropmul),
                        ov.ptr, ov.ld, vptr(pv), vld(pv), None, 0, None if res is None else res.v.ptr,
                        0 if res is None else res.v.ld, rt.stream)
    if out_stats is not None:
        _sum_into(tape, pv if pool else ov, out_stats)
    oa = Act(ov)
    pa = Act(pv) if pool else None
    if fused_bn and not pool and res is None and dropmul is None and tape.need_grad:
        # (the sums buffer is the one this layer's backward reads; cleared by segnb_bn_fwd_fused above)
        oa.producer = (y, coef_buf, tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64), act, slope)

    def backward():
        if oa._g is None and (pa is None or pa.g is None):
            if fused_bn:                     # nothing will clear the forward statistics: not a replayable backward
                stats.zero_()
                tape.unplannable = True
            return
        flat = tape.flat
        dz = tape.view(site + '/dz', N, Ho, Wo, Cp)
        sums = tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64)
        bcoef = tape.small(site + '/bcoef', (3, Cp), torch.float32)
        gp = pa.g if pa is not None else None
        # (two gradient sources -- a tensor with two consumers, linknet.py:41-62's identity branches: both go to the reduction pass,
        # segnb_bn_act_bwd_reduce_add, instead of an add pass first; not beside a pooled gradient)
        og, og2 = tape.sources(oa, gp is None and not oa.g_is_dz)
        # one direct gradient source and nothing else in the way: dz stays in registers (sums-only reduce, then the apply
        # launch recomputes it from the incoming gradient -- segnb_bn_bwd_apply_fused_direct, as ZF_UNET's first convolutions)
        direct = fused_bn and gp is None and res is None and dropmul is None and og is not None and og2 is None
        if oa.g_is_dz:
            # the one consumer's data gradient did the reduction in its epilogue (_data_gradient): the sums are complete and
            # oa.g is the plain gradient, from which the apply launch recomputes dz (the direct form; producer => direct)
            assert direct
            oa.g_is_dz = False
        else:
            if not has_bn and res is None:
                (rt.arm_fork() if forks else None)                # (no BatchNorm: this pass is the last launch before the weight gradient's fork)
            if og2 is not None:
                nv.call('segnb_bn_act_bwd_reduce_add', rt.code, y.ptr, y.ld, N, Ho, Wo, Cp, nv.ptr(coef), act, slope,
                        nv.ptr(dropmul), og.ptr, og.ld, og2.ptr, og2.ld, dz.ptr, dz.ld, nv.ptr(sums),
                        None if res is None else res.v.ptr, 0 if res is None else res.v.ld, rt.stream)
            else:
                nv.call('segnb_bn_act_bwd_reduce', rt.code, y.ptr, y.ld, N, Ho, Wo, Cp, nv.ptr(coef), act, slope,
                        nv.ptr(dropmul), vptr(og), vld(og), vptr(gp), vld(gp), None, 0, None if direct else dz.ptr,
                        0 if direct else dz.ld, nv.ptr(sums), None if res is None else res.v.ptr, 0 if res is None else res.v.ld,
                        rt.stream)
        count = float(N * Ho * Wo)
        dy = dz
        if has_bn:
            gamma = tape.small(site + '/geff', (C,), torch.float32) if abs_form else bn.weight      # (geff: written by the forward)
            if res is not None:             # dz is also the residual branch's gradient: keep it intact
                dy = tape.view(site + '/dy', N, Ho, Wo, Cp)
            if res is None and not abs_form:
                # the apply pass is the LAST launch before the weight gradient's fork below: its event rides on that dispatch
                # (engine.Runtime.arm_fork: no marker packet between the pass and the data gradient on this queue)
                (rt.arm_fork() if forks else None)
            if direct:
                nv.call('segnb_bn_bwd_apply_fused_direct', rt.code, y.ptr, y.ld, N, Ho, Wo, C, Cp, nv.ptr(coef_buf),
                        nv.ptr(sums), nv.ptr(gamma.detach()), nv.ptr(bcoef), nv.ptr(dgamma_target()),
                        nv.ptr(flat.grad_of(bn.bias)), 1, nv.ptr(stats), act, slope, oa.g.ptr, oa.g.ld, dy.ptr, dy.ld,
                        rt.stream)
            elif fused_bn:
                nv.call('segnb_bn_bwd_apply_fused', rt.code, y.ptr, y.ld, N, Ho, Wo, C, Cp, nv.ptr(coef_buf), nv.ptr(sums),
                        nv.ptr(gamma.detach()), nv.ptr(bcoef), nv.ptr(dgamma_target()),
                        nv.ptr(flat.grad_of(bn.bias)), 1, nv.ptr(stats), dz.ptr, dz.ld, dy.ptr, dy.ld, rt.stream)
            else:
                nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, count, nv.ptr(gamma.detach()), nv.ptr(coef_buf),
                        nv.ptr(bcoef), nv.ptr(dgamma_target()), nv.ptr(flat.grad_of(bn.bias)), 1, rt.stream)
                nv.call('segnb_bn_bwd_apply', rt.code, y.ptr, y.ld, N, Ho, Wo, Cp, nv.ptr(coef_buf), nv.ptr(bcoef),
                        dz.ptr, dz.ld, dy.ptr, dy.ld, None, C, rt.stream)
            dgamma_done()
        else:
            gb = flat.grad_of(bias) if bias is not None else None
            tape.defer_bias_grad(sums, C, Cp, gb, count, coef_buf, bcoef)
        if res is not None:
            tape.contribute(res, dz)
        # the weight gradient (and its unpack) only READ x and dy, and nothing reads dW before the end of backward: side
        # stream, beside the dependent chain reduce -> apply -> data gradient (engine.Runtime.fork_side; dy is a buffer
        # of this call site that nothing writes again during this backward)
        side = rt.fork_side() if forks else None
        if side is not None:
            with torch.cuda.stream(side):
                conv.wgrad(xv, dy, flat.grad_of(weight), unpack=False)
        else:
            conv.wgrad(xv, dy, flat.grad_of(weight), unpack=False)
        tape.defer_unpack(conv, xv.H, xv.W, flat.grad_of(weight))
        if x.needs_grad:
            _data_gradient(tape, conv, x, dy, site)

    tape.record(backward)
    return (oa, pa) if pool else oa


def _bn_act_core(tape, x, fields, grads_of, act, slope, tag, out=None, stats_src=None):
    """bn_act over ONE contiguous run of real channels (+ its padding); fields: (gamma, beta, running_mean, running_var,
    num_batches_tracked or None, eps, momentum); grads_of() -> (dgamma, dbeta) fp32 views of the flat gradient buffer.
    stats_src: (table, element offset, row stride) -- the batch statistics of x are ALREADY in that channel range of its concat
    buffer's table (summed slice by slice as the slices were written: conv_unit's out_stats): no pass over x for them."""
    rt, xv = tape.rt, x.v
    site = tape.site(tag)
    tape.consume(x)
    N, H, W, Cp = xv.N, xv.H, xv.W, xv.Cp
    gamma, beta, rm, rv, nbt, eps, mom = fields
    C = gamma.numel()
    stats = tape.small(site + '/stats', (STAT_REPLICAS, 2, Cp), torch.float64)
    coef = tape.small(site + '/coef', (4, Cp), torch.float32)
    fused = tape.fuses_finalize()
    cached = stats_src is not None and fused
    if tape.train and not cached:
        nv.call('segnb_bn_stats', rt.code, xv.ptr, xv.ld, N, H, W, Cp, nv.ptr(stats), rt.stream)
    ov = out if out is not None else tape.view(site + '/a', N, H, W, Cp)
    if cached:
        sums_f = tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64)
        nv.call('segnb_bn_fwd_fused_ld', rt.code, xv.ptr, xv.ld, N, H, W, C, Cp, nv.ptr(stats_src[0], stats_src[1]), stats_src[2],
                nv.ptr(gamma.detach()), nv.ptr(beta.detach()), eps, mom, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), nv.ptr(coef),
                nv.ptr(sums_f), act, slope, None, ov.ptr, ov.ld, rt.stream)
    elif fused:
        sums_f = tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64)
        nv.call('segnb_bn_fwd_fused', rt.code, xv.ptr, xv.ld, N, H, W, C, Cp, nv.ptr(stats), nv.ptr(gamma.detach()),
                nv.ptr(beta.detach()), eps, mom, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), nv.ptr(coef), nv.ptr(sums_f), act,
                slope, None, ov.ptr, ov.ld, None, 0, None, 0, None, 0, rt.stream)
        tape.note_fused_stats(stats)
    else:
        nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * H * W), nv.ptr(gamma.detach()), nv.ptr(beta.detach()),
                eps, mom, nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), 1 if tape.train else 0, nv.ptr(coef), rt.stream)
        nv.call('segnb_bn_act_fwd', rt.code, xv.ptr, xv.ld, N, H, W, Cp, nv.ptr(coef), act, slope, None, ov.ptr, ov.ld,
                None, 0, None, 0, None, 0, rt.stream)
    oa = Act(ov)
    if fused and tape.need_grad and tape.train:
        # the ONE consumer's data gradient may do this layer's BatchNorm-backward reduction in its store pass (_data_gradient:
        # a dense layer's 16 -> prefix data gradient, tiramisu.py:9-20); the sums buffer was cleared by the fused forward above
        oa.producer = (xv, coef, tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64), act, slope)
        # ... or never store that gradient at all: sums launch now, recompute + apply in this layer's backward (nothing else to
        # clear there: the statistics are the concat buffer's cached ones)
        oa.lazy_ok = cached

    def backward():
        if oa.lazy_dgrad is None and oa.g is None:
            if fused and not cached:
                stats.zero_()
                tape.unplannable = True
            return
        flat = tape.flat
        sums = tape.small(site + '/sums', (STAT_REPLICAS, 2, Cp), torch.float64)
        bcoef = tape.small(site + '/bcoef', (3, Cp), torch.float32)
        if oa.lazy_dgrad is not None:
            conv2, dyv = oa.lazy_dgrad
            oa.lazy_dgrad, oa.g_is_dz = None, False
            acc = x.needs_grad and x.g is not None
            dst = x.g if acc else tape.view(site + '/dz', N, H, W, Cp)
            ep = nv.BnApplyEpilogue(xv.ptr, xv.ld, nv.ptr(coef), nv.ptr(sums), nv.ptr(gamma.detach()), C, float(N * H * W),
                                    nv.ptr(bcoef), nv.ptr(grads_of()[0]), nv.ptr(grads_of()[1]), act, slope, dst.ptr, dst.ld,
                                    1 if acc else 0)
            conv2.dgrad_bnapply(dyv, H, W, ep)
            if not acc:
                tape.contribute(x, dst)
            return
        if fused:
            # ONE gradient source, no dropout, no residual: dz never goes to memory -- a sums-only reduction (done by the
            # consumer's data gradient where a fused kernel serves it: oa.g_is_dz), then the apply launch recomputes
            # dz = act'(z) * g from the incoming gradient (the *_direct forms, as ZF_UNET's first convolutions) and, when the
            # input has other consumers whose gradients are already in x.g, adds its result there (no segnb_add pass)
            if oa.g_is_dz:
                oa.g_is_dz = False
            else:
                nv.call('segnb_bn_act_bwd_reduce', rt.code, xv.ptr, xv.ld, N, H, W, Cp, nv.ptr(coef), act, slope, None,
                        oa.g.ptr, oa.g.ld, None, 0, None, 0, None, 0, nv.ptr(sums), None, 0, rt.stream)
            fargs = (rt.code, xv.ptr, xv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums), nv.ptr(gamma.detach()),
                     nv.ptr(bcoef), nv.ptr(grads_of()[0]), nv.ptr(grads_of()[1]), 1, None if cached else nv.ptr(stats),
                     act, slope, oa.g.ptr, oa.g.ld)
            if x.needs_grad and x.g is not None:
                nv.call('segnb_bn_bwd_apply_fused_direct_acc', *(fargs + (x.g.ptr, x.g.ld, rt.stream)))
            else:
                dz = tape.view(site + '/dz', N, H, W, Cp)
                nv.call('segnb_bn_bwd_apply_fused_direct', *(fargs + (dz.ptr, dz.ld, rt.stream)))
                tape.contribute(x, dz)
            return
        dz = tape.view(site + '/dz', N, H, W, Cp)
        nv.call('segnb_bn_act_bwd_reduce', rt.code, xv.ptr, xv.ld, N, H, W, Cp, nv.ptr(coef), act, slope, None,
                oa.g.ptr, oa.g.ld, None, 0, None, 0, dz.ptr, dz.ld, nv.ptr(sums), None, 0, rt.stream)
        nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, float(N * H * W), nv.ptr(gamma.detach()), nv.ptr(coef),
                nv.ptr(bcoef), nv.ptr(grads_of()[0]), nv.ptr(grads_of()[1]), 1, rt.stream)
        nv.call('segnb_bn_bwd_apply', rt.code, xv.ptr, xv.ld, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), dz.ptr, dz.ld,
                dz.ptr, dz.ld, None, C, rt.stream)
        tape.contribute(x, dz)

    tape.record(backward)
    return oa


def bn_act(tape, x, bn, act=nv.ACT_RELU, slope=0.01, tag='bnact', segs=None, stats_src=None):
    """PRE-activation BatchNorm + activation of an arbitrary tensor (tiramisu.py:12-13, 50-51).

    segs: [(real, padded), ...] when the tensor is a concat of padded slices whose real channels are the BatchNorm's
    consecutive features (a dense block with a growth rate that is not a multiple of 8: tiramisu.py:187-191) -- BatchNorm is
    per channel, so the layer runs as one launch set per slice on sub-views of the parameters; num_batches_tracked is
    advanced by the first slice only."""
    if segs is None or len(segs) == 1:
        flat = tape.flat
        return _bn_act_core(tape, x, _bn_fields(bn), lambda: (flat.grad_of(bn.weight), flat.grad_of(bn.bias)), act, slope, tag,
                            stats_src=stats_src)
    gamma, beta, rm, rv, nbt, eps, mom = _bn_fields(bn)
    xv = x.v
    assert sum(p for _, p in segs) == xv.Cp and sum(r for r, _ in segs) == gamma.numel(), (segs, xv.Cp, gamma.numel())
    ov = tape.view(tape.site(tag) + '/a', xv.N, xv.H, xv.W, xv.Cp)
    oa = Act(ov)
    parts, roff, poff = [], 0, 0
    for k, (real, padded) in enumerate(segs):
        xs = Act(xv.slice(poff, padded), needs_grad=x.needs_grad)
        xs.g = x.g.slice(poff, padded) if x.g is not None else None
        sl = slice(roff, roff + real)
        fields = (gamma[sl], beta[sl], rm[sl], rv[sl], nbt if k == 0 else None, eps, mom)

        def grads_of(sl=sl):
            flat = tape.flat
            return flat.grad_of(bn.weight)[sl], flat.grad_of(bn.bias)[sl]
        part = _bn_act_core(tape, xs, fields, grads_of, act, slope, '%s.s%d' % (tag, k), out=ov.slice(poff, padded),
                            stats_src=None if stats_src is None else (stats_src[0], stats_src[1] + poff, stats_src[2]))
        parts.append((part, poff, padded))
        roff += real
        poff += padded

    def bind():
        for part, po, pd in parts:
            part.g = oa.g.slice(po, pd) if oa.g is not None else None
    # the slices' backward closures were recorded above and run AFTER this one (the tape runs in reverse): bind first
    tape.record(bind)
    return oa


def maxpool(tape, x, k, stride, pad, tag='pool'):
    rt, xv = tape.rt, x.v
    site = tape.site(tag)
    tape.consume(x)
    Ho, Wo = (xv.H + 2 * pad - k) // stride + 1, (xv.W + 2 * pad - k) // stride + 1
    ov = tape.view(site + '/o', xv.N, Ho, Wo, xv.Cp)
    # argmax positions recorded by the forward for the backward (a re-scanning backward costs 36 loads per pixel)
    idx = tape.small(site + '/idx', (xv.N, Ho, Wo, xv.Cp), torch.uint8) if tape.need_grad and x.needs_grad else None
    nv.call('segnb_maxpool_fwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, xv.Cp, k, stride, pad, ov.ptr, ov.ld,
            nv.ptr(idx), rt.stream)
    oa = Act(ov)

    def backward():
        if oa._g is None or not x.needs_grad:
            return
        dx = tape.view(site + '/dx', xv.N, xv.H, xv.W, xv.Cp)
        # (a held-back second gradient of the pooled tensor -- Tape.lazy_add -- is added on the way: segnb_maxpool_bwd_add)
        g1, g2 = tape.sources(oa, idx is not None)
        if g2 is not None:
            nv.call('segnb_maxpool_bwd_add', rt.code, xv.ptr, xv.ld, g1.ptr, g1.ld, g2.ptr, g2.ld, xv.N, xv.H, xv.W, xv.Cp, k, stride,
                    pad, dx.ptr, dx.ld, nv.ptr(idx), rt.stream)
        else:
            nv.call('segnb_maxpool_bwd', rt.code, xv.ptr, xv.ld, g1.ptr, g1.ld, xv.N, xv.H, xv.W, xv.Cp, k, stride,
                    pad, dx.ptr, dx.ld, nv.ptr(idx), rt.stream)
        tape.contribute(x, dx)

    tape.record(backward)
    return oa


def upsample_bilinear2x(tape, x, tag='up'):
    """nn.Upsample(scale_factor=2, mode='bilinear') (unet16.py:43: DecoderBlock with is_deconv=False)."""
    rt, xv = tape.rt, x.v
    site = tape.site(tag)
    tape.consume(x)
    ov = tape.view(site + '/o', xv.N, 2 * xv.H, 2 * xv.W, xv.Cp)
    nv.call('segnb_upsample_bilinear2x_fwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, xv.Cp, ov.ptr, ov.ld, rt.stream)
    oa = Act(ov)

    def backward():
        if oa.g is None or not x.needs_grad:
            return
        dx = tape.view(site + '/dx', xv.N, xv.H, xv.W, xv.Cp)
        nv.call('segnb_upsample_bilinear2x_bwd', rt.code, oa.g.ptr, oa.g.ld, xv.N, xv.H, xv.W, xv.Cp, dx.ptr, dx.ld, rt.stream)
        tape.contribute(x, dx)

    tape.record(backward)
    return oa


def add(tape, a, b, tag='add'):
    """out = a + b (linknet.py:77-79); both inputs receive the output gradient unchanged."""
    rt, av, bv = tape.rt, a.v, b.v
    site = tape.site(tag)
    tape.consume(a, b)
    ov = tape.view(site + '/o', av.N, av.H, av.W, av.Cp)
    nv.call('segnb_add', rt.code, av.ptr, av.ld, bv.ptr, bv.ld, ov.ptr, ov.ld, av.N, av.H, av.W, av.Cp, rt.stream)
    oa = Act(ov)

    def backward():
        if oa.g is None:
            return
        # two consumers of the same buffer: the second contribution must not alias the first adopter
        tape.contribute(a, oa.g)
        if b.g is None and b.needs_grad:
            gb = tape.view(site + '/gb', av.N, av.H, av.W, av.Cp)
            rt.copy_view(oa.g, gb)
            b.g = gb
        else:
            tape.contribute(b, oa.g)

    tape.record(backward)
    return oa


def concat(tape, pieces, cat_view):
    """pieces: [(Act, channel offset)] already WRITTEN into slices of cat_view (zero-copy torch.cat)."""
    ca = Act(cat_view)
    tape.consume(*[act for act, _ in pieces])

    def backward():
        if ca.g is None:
            return
        for act, off in pieces:
            tape.contribute(act, ca.g.slice(off, act.v.Cp))

    tape.record(backward)
    return ca


def head_1x1(tape, x, weight, bias, dlogits_ref, tag='head'):
    """1x1 classifier -> fp32 NCHW logits (unet16.py:111, tiramisu.py:162-164)."""
    rt, xv = tape.rt, x.v
    site = tape.site(tag)
    tape.consume(x)
    K, C = weight.shape[0], weight.shape[1]
    logits = tape.cached((site, xv.N, xv.H, xv.W), lambda: torch.zeros((xv.N, K, xv.H, xv.W), dtype=torch.float32,
                                                                       device=rt.device))
    nv.call('segnb_head_fwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, C, nv.ptr(weight.detach()),
            nv.ptr(bias.detach()), K, nv.ptr(logits), rt.stream)

    def backward():
        flat = tape.flat
        da = tape.view(site + '/da', xv.N, xv.H, xv.W, xv.Cp)
        mask = _head_mask(tape, x, C, K, 1, 1)
        if mask is not None:
            sums, act, slope = mask
            nv.call('segnb_head_conv_bwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, C, xv.Cp, nv.ptr(weight.detach()), 1, 1, 0, K,
                    nv.ptr(dlogits_ref[0]), act, slope, da.ptr, da.ld, nv.ptr(flat.grad_of(weight)), nv.ptr(flat.grad_of(bias)),
                    nv.ptr(sums), rt.stream)
            x.g_is_dz = True
        else:
            nv.call('segnb_head_bwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, C, xv.Cp, nv.ptr(weight.detach()), K,
                    nv.ptr(dlogits_ref[0]), da.ptr, da.ld, nv.ptr(flat.grad_of(weight)), nv.ptr(flat.grad_of(bias)),
                    rt.stream)
        tape.contribute(x, da)

    tape.record(backward)
    return logits


def _head_mask(tape, x, C, K, kh, kw):
    """(sums, act, slope) when this head is the ONLY consumer of a convolution + activation without BatchNorm (unet16.py:110-111
    dec1 -> final, linknet.py:58-62 finalrelu2 -> finalconv3) and the head kernels serve the shape: the head's backward then writes
    that layer's dz and sums it (Act.producer with coef None), instead of a gradient the layer would mask in a pass of its own."""
    pr = x.producer
    if (pr is None or pr[1] is not None or not Tape.fold_head_mask or x.consumers != 1 or x.g is not None or not x.needs_grad
            or not nv.query('segnb_head_conv_ok', C, K, kh, kw)):
        return None
    return pr[2], pr[3], pr[4]


def head_conv(tape, x, weight, bias, pad, dlogits_ref, tag='headconv'):
    """A classifier that is a small stride-1 convolution (linknet.py:62: Conv2d(32, num_classes, 2, padding=1)) -> fp32 NCHW logits,
    on the head kernels (segnb_head_conv_fwd / _bwd; check segnb_head_conv_ok first): the parameter is read and its gradient written
    in its own layout, no packed copies."""
    rt, xv = tape.rt, x.v
    site = tape.site(tag)
    tape.consume(x)
    K, C, kh, kw = weight.shape
    Ho, Wo = xv.H + 2 * pad - kh + 1, xv.W + 2 * pad - kw + 1
    logits = tape.cached((site, xv.N, xv.H, xv.W), lambda: torch.zeros((xv.N, K, Ho, Wo), dtype=torch.float32, device=rt.device))
    nv.call('segnb_head_conv_fwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, C, nv.ptr(weight.detach()), kh, kw, pad,
            nv.ptr(bias.detach()) if bias is not None else None, K, nv.ptr(logits), rt.stream)

    def backward():
        flat = tape.flat
        da = tape.view(site + '/da', xv.N, xv.H, xv.W, xv.Cp) if x.needs_grad else None
        mask = _head_mask(tape, x, C, K, kh, kw)
        sums, act, slope = mask if mask is not None else (None, -1, 0.0)
        nv.call('segnb_head_conv_bwd', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, C, xv.Cp, nv.ptr(weight.detach()), kh, kw, pad, K,
                nv.ptr(dlogits_ref[0]), act, slope, vptr(da), vld(da), nv.ptr(flat.grad_of(weight)),
                nv.ptr(flat.grad_of(bias)) if bias is not None else None, nv.ptr(sums), rt.stream)
        if mask is not None:
            x.g_is_dz = True
        if da is not None:
            tape.contribute(x, da)

    tape.record(backward)
    return logits


def head_from_act(tape, x, K, dlogits_ref, tag='headconv'):
    """An activation tensor whose first K channels ARE the logits (head that is a general conv, linknet.py:62):
    NHWC -> fp32 NCHW forward, fp32 NCHW gradient -> NHWC backward."""
    rt, xv = tape.rt, x.v
    site = tape.site(tag)
    tape.consume(x)
    logits = tape.cached((site, xv.N, xv.H, xv.W), lambda: torch.zeros((xv.N, K, xv.H, xv.W), dtype=torch.float32,
                                                                       device=rt.device))
    nv.call('segnb_nhwc_to_nchw_f32', rt.code, xv.ptr, xv.ld, xv.N, xv.H, xv.W, K, nv.ptr(logits), rt.stream)

    def backward():
        g = tape.view(site + '/g', xv.N, xv.H, xv.W, xv.Cp)
        nv.call('segnb_pack_input_nchw', nv.ptr(dlogits_ref[0]), xv.N, K, xv.H, xv.W, g.ptr, rt.code, xv.Cp, g.ld,
                rt.stream)
        tape.contribute(x, g)

    tape.record(backward)
    return logits


# ------------------------------------------------------------------------------------------------------------
# nn.Module base
# ------------------------------------------------------------------------------------------------------------
class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.net = net
        ctx.nparams = len(params)
        out = net._run(x, True)
        ctx.generation = net._tape.generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.generation != ctx.net._tape.generation:
            raise RuntimeError('%s: another forward ran on this model since the forward being differentiated; run '
                               'backward before the next forward (the tape keeps ONE set of activation buffers)'
                               % type(ctx.net).__name__)
        ctx.net._run_backward(dlogits.contiguous().float())
        return (None, None) + (None,) * ctx.nparams


class HipNet(nn.Module):
    """Base of the executor-driven models: subclasses implement ``_build(tape, x_act, dlogits_ref) -> logits``."""

    # share (%) of the CUs the weight gradients (side stream) split their pixels for; None = the library default (half, the
    # other half left to the dependent chain).  A model whose side stream is the longer one overrides it (UNet16).
    wg_cu_pct = None

    def _init_engine(self, in_channels):
        self.compute_dtype = 'bf16'
        self._tape = None
        self._in_channels = in_channels
        self.input_norm = InputNorm()           # for uint8 NHWC batches (lib/augmentations.py:452-460 on the device)

    def set_compute_dtype(self, dtype):
        if dtype not in ('bf16', 'f32'):
            raise ValueError("compute dtype must be 'bf16' or 'f32'")
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._tape = None
        return self

    def _check_input(self, x):
        cdim = 3 if x.dtype == torch.uint8 else 1
        if x.dim() != 4 or x.shape[cdim] != self._in_channels:
            raise ValueError('expected input [N, %d, H, W] (float) or [N, H, W, %d] (uint8), got %s %s'
                             % (self._in_channels, self._in_channels, x.dtype, tuple(x.shape)))

    def forward(self, x):
        """x: float32 [N, C, H, W], or uint8 [N, H, W, C] normalised on the device with ``self.input_norm``."""
        if x.dtype == torch.uint8:              # the shape checks of the subclasses speak NCHW
            self._check_input(torch.empty((x.shape[0], x.shape[3], x.shape[1], x.shape[2]), device='meta'))
            x = x.detach().contiguous()
        else:
            self._check_input(x)
            x = x.detach().contiguous().float()
        if self._tape is None or self._tape.rt.device != x.device:
            self._tape = Tape(self, x.device, self.compute_dtype)
        if torch.is_grad_enabled():
            flat = getattr(getattr(self, '_tape', None), 'flat', None)
            params = flat.param_list() if flat is not None else list(self.parameters())
            if any(p.requires_grad for p in params):
                return _NetFn.apply(self, x, *params)
        return self._run(x, False)

    # ---- recorded launch lists (segnb_plan_*: include/segnb_hip.h) --------------------------------------------------------
    # The graphs are static: the SECOND step of a (geometry, mode) runs the Python build once more while the library records
    # every ABI call it makes (the first one allocated buffers and packed weights with calls that are not replayable); from
    # the third step on the forward and the backward are one segnb_plan_run each -- 3-4 us of host time per launch instead
    # of the 10-14 us of a ctypes call, which is what bounds FCDenseNet103's ~2250 launches of ~10 us.  Whatever varies
    # between steps enters through persistent buffers: the input batch and the logits gradient are copied into them, the
    # dropout pools are redrawn in Tape.begin, BatchNorm statistics are cleared by the kernels that consume them.
    use_cplan = os.environ.get('SEGNB_CPLAN', '1') != '0'

    def _plan_key(self, x, need_grad):
        from . import engine
        tape = self._tape
        rt = tape.rt
        if not self.use_cplan or rt.device.type != 'cuda' or engine.TIMER is not None:
            return None
        side = rt.side_stream()
        return (tuple(x.shape), x.dtype, bool(self.training), bool(need_grad), rt.stream, side.cuda_stream if side is not None else 0,
                tape.flat.flat_p.data_ptr(), tape.flat.flat_g.data_ptr(), tuple(b.data_ptr() for b in tape.flat.buffer_list()),
                tuple(sorted((p, pool['used']) for p, pool in tape._drop_pools.items())) if self.training else (),
                tape._ready_hook() is not None)

    @staticmethod
    def _plan_drop(ent):
        for h in [ent.get('fwd')] + [h for h, _ in (ent.get('bwd') or [])]:
            if h:
                nv.call('segnb_plan_destroy', h)
        ent['fwd'] = ent['bwd'] = None

    def _guards(self):
        g = getattr(self, '_replay_guards', None)
        if g is None:
            from .engine import ReplayGuard
            g = self._replay_guards = (ReplayGuard(type(self).__name__ + ' forward'), ReplayGuard(type(self).__name__ + ' backward'))
        return g

    def _run(self, x, need_grad):
        # (SEGNB_REPLAY_GUARD=1: a replayed forward must execute the launches of the forward that recorded its list)
        # the census starts where the two paths part (the list look-up): Tape.begin -- dropout pools, the weight pack when the
        # parameters changed -- the gradient-buffer clear and the input pack are the same host code either way
        self._guard_mode, self._guard_active = (None, None), False
        out = self._run_(x, need_grad)
        self._guards()[0].end(self._guard_active, *self._guard_mode)
        return out

    def _run_(self, x, need_grad):
        tape = self._tape
        tape.begin(self.training, need_grad)
        if self.training and need_grad:
            tape.flat.prezero(tape.rt)          # the coming backward's gradient-buffer clear, on the side stream beside the forward
        if x.dtype == torch.uint8:
            N, H, W, C = x.shape
        else:
            N, C, H, W = x.shape
        key = self._plan_key(x, need_grad)
        ent, recording = None, False
        self._plan_live = None
        # the batch enters through ONE launch outside the recorded list (it reads the caller's tensor, whatever its address):
        # NCHW fp32 / NHWC uint8 -> the padded NHWC buffer the list's first convolution reads
        cin_p = cp.pad8(C)
        xin = tape.view('input', N, H, W, cin_p)
        pack_input(tape.rt, x, xin, getattr(self, 'input_norm', None))
        if key is not None:
            ent = tape.plans.get(key)
            self._guard_active = self._guards()[0].begin()
            if ent is None:
                ent = tape.plans[key] = {'state': 'seen'}                  # first step of this key: eager
            else:
                if ent['state'] == 'ready':
                    self._guard_mode = (key, 'replay')
                    nv.call('segnb_plan_run', ent['fwd'])
                    tape.back = []
                    tape.fused_stats = ent['fused_stats']
                    tape.stats_pending = bool(tape.fused_stats)
                    self._plan_live = ent if need_grad else None
                    return ent['logits'].clone()
                if ent['state'] == 'seen' or (ent['state'] == 'fwd' and need_grad):
                    self._plan_drop(ent)                                    # (a recorded forward whose backward never ran)
                    nv.plan_record_begin()
                    recording = True
        self._dlogits = [None]
        tape.unplannable = False
        try:
            logits = self._build(tape, Act(xin, needs_grad=False), self._dlogits)
        except BaseException:
            if recording:                      # close the abandoned recording: this key stays eager
                nv.plan_record_abort()
                ent['state'] = 'eager'
            raise
        tape.stats_pending = bool(tape.fused_stats)
        if recording:
            handle, nops = nv.plan_record_end()
            if handle is None:
                ent['state'] = 'eager'                                     # not replayable: remembered
            else:
                ent.update(fwd=handle, logits=logits, nfwd=nops, state='fwd' if need_grad else 'ready',
                           fused_stats=list(tape.fused_stats))
                self._guard_mode = (key, 'record')
                self._plan_live = ent if need_grad else None
        return logits.clone()

    def _run_backward(self, dlogits):
        self._guard_mode, self._guard_active = (None, None), False
        from . import engine as _engine
        try:
            out = self._run_backward_(dlogits)
        finally:
            _engine.DW_OVERWRITE = False
        self._guards()[1].end(self._guard_active, *self._guard_mode)
        return out

    def _run_backward_(self, dlogits):
        tape = self._tape
        if not tape.train and any(isinstance(m, nn.modules.batchnorm._BatchNorm) or type(m).__name__ == 'InPlaceABN'
                                  for m in tape.flat.module_list()):
            raise RuntimeError('backward through an eval-mode forward is not supported: BatchNorm gradients are '
                               'implemented for training mode (call model.train(), or run the forward under no_grad)')
        ent = getattr(self, '_plan_live', None)
        self._plan_live = None
        acc = tape.flat.begin_backward()
        # directly delivered weight gradients are STORED when the gradient buffer was just cleared, ADDED when gradients accumulate in
        # place (engine.DW_OVERWRITE); the recorded backward lists carry the flag they were recorded under
        from . import engine as _engine
        _engine.DW_OVERWRITE = not acc
        # a list recorded with STORES replayed by a backward that accumulates on top of earlier gradients: the fresh gradient goes to a
        # cleared buffer and the earlier one is added back (rare: lib/train_utils.find_optimal_lr never zeroes; a list recorded with
        # ADDS is right either way)
        stash = None
        if ent is not None and ent['state'] == 'ready' and acc and ent.get('acc') is False:
            stash = tape.flat.flat_g.clone()
            tape.flat.flat_g.zero_()
        if ent is not None:
            din = ent.get('dlogits_in')
            if din is None:
                din = ent['dlogits_in'] = torch.empty_like(dlogits)
            din.copy_(dlogits)                                              # (autograd hands over a new tensor every step)
            dlogits = din
        self._guard_active = self._guards()[1].begin() if ent is not None else False
        if ent is not None and ent['state'] == 'ready':
            self._guard_mode = (id(ent), 'replay')
            for handle, cut in ent['bwd']:
                nv.call('segnb_plan_run', handle)
                if cut is not None:          # host work between two segments: partial unpack + the data-parallel hook
                    table, lo, on_side = cut
                    side = tape.rt.side_stream() if on_side else None
                    if table is not None:
                        if side is not None:
                            with torch.cuda.stream(side):
                                table.run()
                        else:
                            table.run()
                    hook = tape._ready_hook()
                    if hook is not None:
                        hook(tape.flat, lo, (side,) if side is not None else ())
            tape.rt._side_busy = False
            tape.stats_pending = False
            if ent['unpack'] is not None:
                ent['unpack'].run()
        else:
            self._dlogits[0] = dlogits
            recording = ent is not None and ent['state'] == 'fwd'
            segs, around = [], None
            paused = [False]          # True while a cut's hook runs OUTSIDE the recording (nothing is open to abort then)
            if recording:
                nv.plan_record_begin()

                def around(do):
                    handle, nops = nv.plan_record_end()
                    paused[0] = True
                    segs.append([handle, nops, None])          # the ended segment is owned by `segs` before the hook can raise
                    segs[-1][2] = do()
                    nv.plan_record_begin()
                    paused[0] = False
            try:
                tape.run_closures(around, join=False)
            except BaseException:
                # (KeyboardInterrupt / SystemExit too: an open recording would swallow every later ABI call of this thread and
                # make segnb_plan_run / segnb_tune refuse -- a validation pass in a Ctrl-C handler would hit that)
                if recording:
                    if not paused[0]:
                        nv.plan_record_abort()
                    for h, _, _ in segs:
                        if h is not None:
                            nv.call('segnb_plan_destroy', h)
                    self._plan_drop(ent)
                    ent['state'] = 'eager'
                raise
            if recording:
                handle, nops = nv.plan_record_end()
                segs.append((handle, nops, None))
                if any(h is None for h, _, _ in segs) or getattr(tape, 'unplannable', False):
                    for h, _, _ in segs:
                        if h is not None:
                            nv.call('segnb_plan_destroy', h)
                    self._plan_drop(ent)
                    ent['state'] = 'eager'
                else:
                    ent.update(bwd=[(h, c) for h, _, c in segs], nbwd=sum(n for _, n, _ in segs), state='ready', acc=acc)
                    self._guard_mode = (id(ent), 'record')
            table = tape.finish()                                           # (7x7 / strided jobs take host tap arrays: eager)
            if recording and ent['state'] == 'ready':
                ent['unpack'] = table
        if stash is not None:
            tape.flat.flat_g.add_(stash)
        hook = getattr(self, '_grad_sync_hook', None)
        if hook is not None:
            hook(tape.flat)
        tape.flat.publish_grads(acc)
