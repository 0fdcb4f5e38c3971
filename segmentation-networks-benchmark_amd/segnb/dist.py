"""Pure data parallelism over the GPUs of one node: one process per GPU, torch.distributed with backend
"nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for tests.

The reference is single-GPU (no DataParallel/DDP anywhere on its main path), so the exchange is
new: per step
  * ONE SUM all-reduce of the flat fp32 gradient buffer (bucketed, issued on a side stream as soon as the
    backward plan has finished the bucket's layers).  SUM, not mean: the reference seeds backward with
    batch_size * loss (torch_train.py:187-188), so per-GPU gradients are already sums over local samples.
  * one 8-double all-reduce of the loss kernel's global sums, so Jaccard/Dice (and the reported loss) are
    computed over the GLOBAL batch exactly as the reference computes them over its whole batch tensor
    (lib/losses.py:39-42).
BatchNorm statistics stay per GPU (what nn.BatchNorm2d does under any torch DP; InPlaceABNSync is unused).
"""
import os

import torch
import torch.distributed as td

from . import _native as nv
from . import seglosses


def world():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    # a 1-rank group is only useful to exercise the collective path on a single GPU (SEGNB_DP_FORCE=1)
    if td.is_initialized() or (ws <= 1 and not os.environ.get('SEGNB_DP_FORCE')) or 'RANK' not in os.environ:
        return
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    td.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)


_conv_cu_pct = 100        # what segnb_tune('conv_cu_pct') was last set to by this module (process-global knob)


class DataParallel(object):
    """Attach gradient / loss-sum synchronisation to a segnb-engine model (in place; returns the model).

    bucket_bytes: all-reduce granularity.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): large
    buckets keep every link busy; the default 32 MiB gives ZF_UNET (125.8 MB of fp32 gradients) 4 buckets.
    """

    def __init__(self, model, bucket_bytes=32 << 20, wire_dtype='f32'):
        """wire_dtype: 'f32' (default: the exchange is exact, the sum of the ranks' fp32 gradients) or 'bf16' -- each
        bucket is rounded to bf16, summed by the collective in bf16 and widened back into the flat fp32 buffer: half the
        bytes per xGMI link (62.9 instead of 125.8 MB per ZF_UNET step), gradients exact to 8 bits."""
        if wire_dtype not in ('f32', 'bf16'):
            raise ValueError("wire_dtype must be 'f32' or 'bf16'")
        self.model = model
        self.wire_dtype = wire_dtype
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.ws = world()
        self.active = self.ws > 1 or (bool(os.environ.get('SEGNB_DP_FORCE')) and td.is_initialized())
        # every backward of a data-parallel job notes whether it accumulates on top of earlier gradients
        # (FlatParams.begin_backward reads this flag through its module): set HERE, before the first backward, not
        # lazily in the middle of one
        model._dp_track_accumulation = self.active
        self._synced = False
        self._comm_stream = None
        model._grad_sync_hook = self.sync_grads
        model._grad_ready_hook = self.grads_ready
        self._pending = []
        self._done_upto = None
        self._fused_opt = None
        self.reserved_cus = 0
        self.trace = None              # {'first_bucket': event, 'backward_end': event} when trace_overlap() was called
        if self.active:
            seglosses.DataParallelHooks.sums_allreduce = self._allreduce_sums
            seglosses.DataParallelHooks.grad_scale = float(self.ws)
            self._reserve_cus()

    def _reserve_cus(self):
        """RCCL's all-reduce kernels run BESIDE the backward; the persistent convolution kernels size their grids for every
        CU of the chip (one block per CU, resident for the whole launch).  SEGNB_DP_RESERVE_CUS = k > 0 sizes those grids for
        CUs - k when world > 1.  Default 0: shrinking a grid does not reserve a CU for anybody (there is no CU mask), so it is
        a certain loss of convolution throughput for a benefit nobody has measured -- the knob exists so that the
        first 8-GPU run can sweep it.  The previous value is restored by detach()."""
        global _conv_cu_pct
        k = int(os.environ.get('SEGNB_DP_RESERVE_CUS', '0'))
        if k <= 0:
            return
        try:
            cus = int(nv.query('segnb_device_cus'))
        except Exception:
            cus = 0
        if cus <= k:
            return
        pct = max(10, (cus - k) * 100 // cus)
        self._restore_conv_pct = _conv_cu_pct
        nv.call('segnb_tune', b'conv_cu_pct', pct)
        _conv_cu_pct = pct
        self.reserved_cus = cus - cus * pct // 100

    def trace_overlap(self):
        """Record a HIP event in front of the FIRST gradient bucket's all-reduce (communication stream) and one at the end
        of backward (compute stream) for the next backward: tests assert that the first collective is enqueued -- and can
        start -- before the backward has finished (tests/test_zf_unet_gpu.py)."""
        self.trace = {}
        return self

    def fuse_optimizer(self, optimizer):
        """Fold the optimizer step into the all-reduce epilogue: each bucket's slice of the flat
        parameter buffer is updated on the communication stream right behind that bucket's all-reduce, beside the rest of
        backward, and ``optimizer.step()`` (torch_train.py:190) finds the work done.  For the segnb.optim classes on their
        one-launch path (one param group holding every parameter of the model, no momentum / weight decay); anything
        else -- and a backward that meets foreign .grad tensors or accumulates -- keeps the ordinary step.
        CONTRACT ("optimizer in the all-reduce epilogue"): the parameters are already updated when
        backward() returns -- skipping ``optimizer.step()`` (a NaN guard), clipping or editing gradients between
        backward() and step() then has NO effect on that step, and the learning rate is the one set when the bucket is
        launched, i.e. during backward (the reference changes it between epochs only).  A finished update that no
        ``optimizer.step()`` consumed is reported with a warning at the next backward."""
        if not hasattr(optimizer, 'fusable_group') or not hasattr(optimizer, 'step_range'):
            raise TypeError('fuse_optimizer needs a segnb.optim optimizer (SGD, RMSprop, Adam)')
        self._fused_opt = optimizer
        return self

    def _fused_group(self, flat):
        """Decided ONCE per backward, when its first bucket (the one that ends at flat.total) is launched, and kept
        for the remaining buckets: a backward is either stepped bucket by bucket from the first one on, or not at all
        (a later bucket with first=False behind an unfused first one double-stepped SGD / broke Adam's counter)."""
        opt = self._fused_opt
        if opt is None or not self.active:
            return None
        if not getattr(flat, 'accumulation_tracked', False) or not getattr(flat, 'fresh_backward', False):
            return None
        return opt.fusable_group(flat)

    def detach(self):
        """Undo __init__: remove the model's hooks and the loss-path hooks (a process that goes on to run the same
        model outside the job -- validation on rank 0, a second DataParallel -- must not keep all-reducing)."""
        for name in ('_grad_sync_hook', '_grad_ready_hook'):
            if getattr(self.model, name, None) is not None:
                delattr(self.model, name)
        if self.active:
            seglosses.DataParallelHooks.reset()
        if getattr(self, '_restore_conv_pct', None) is not None:
            global _conv_cu_pct
            nv.call('segnb_tune', b'conv_cu_pct', self._restore_conv_pct)
            _conv_cu_pct, self._restore_conv_pct, self.reserved_cus = self._restore_conv_pct, None, 0
        self.active = False
        self.model._dp_track_accumulation = False

    def __call__(self, *a, **k):
        return self.model(*a, **k)

    # ---- parameters ------------------------------------------------------------------------------------
    def broadcast_parameters(self, flat):
        if self.active and not self._synced:
            td.broadcast(flat.flat_p, src=0)
            for b in self.model.buffers():
                td.broadcast(b, src=0)
            flat.version += 1
        self._synced = True

    # ---- loss sums -------------------------------------------------------------------------------------
    def _allreduce_sums(self, sums):
        td.all_reduce(sums, op=td.ReduceOp.SUM)

    # ---- gradients -------------------------------------------------------------------------------------
    def _stream(self, flat):
        if not flat.flat_g.is_cuda:
            return None
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=flat.flat_g.device)
        return self._comm_stream

    def grads_ready(self, flat, lo, producers=()):
        """Called by the backward plan when every gradient at flat offset >= lo is final (the plan runs
        decoder -> encoder, i.e. from the END of the flat buffer towards its start).  Launches the
        all-reduce of every full bucket that became ready, on the communication stream, which waits for the
        current stream and for `producers` (other streams that wrote part of those gradients)."""
        if not self.active:
            return
        self._producers = tuple(producers)
        hi = flat.total if self._done_upto is None else self._done_upto
        while hi - lo >= self.bucket_elems or (lo == 0 and hi > 0):
            start = max(lo, hi - self.bucket_elems) if lo > 0 else max(0, hi - self.bucket_elems)
            self._launch(flat, start, hi)
            hi = start
            if hi == 0:
                break
        self._done_upto = hi

    def _launch(self, flat, start, end):
        chunk = flat.flat_g[start:end]
        cs = self._stream(flat)
        first = end == flat.total
        if first:
            self._fused_cover = 0
            self._group_this_backward = self._fused_group(flat)
        group = getattr(self, '_group_this_backward', None)
        bf16 = self.wire_dtype == 'bf16'
        if cs is None:
            if bf16:
                wire = chunk.to(torch.bfloat16)
                td.all_reduce(wire, op=td.ReduceOp.SUM)
                chunk.copy_(wire)
                work = None
            else:
                work = td.all_reduce(chunk, op=td.ReduceOp.SUM, async_op=True)
            if group is None:
                if work is not None:
                    self._pending.append(work)
                return
            if work is not None:
                work.wait()
            self._fused_opt.step_range(flat, group, start, end, first)
            self._fused_cover += end - start
            return
        cs.wait_stream(torch.cuda.current_stream(flat.flat_g.device))
        for ps in getattr(self, '_producers', ()):
            cs.wait_stream(ps)
        with torch.cuda.stream(cs):
            if self.trace is not None and first:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(cs)
                self.trace['first_bucket'] = ev
            if bf16:
                wire = chunk.to(torch.bfloat16)       # (allocated on the communication stream: reused in stream order)
                td.all_reduce(wire, op=td.ReduceOp.SUM)
                chunk.copy_(wire)
            else:
                td.all_reduce(chunk, op=td.ReduceOp.SUM)
            if group is not None:
                self._fused_opt.step_range(flat, group, start, end, first)
                self._fused_cover += end - start

    def sync_grads(self, flat):
        """End of backward: reduce whatever is left, then make the compute stream wait for the collectives."""
        if not self.active:
            return
        if getattr(flat, 'accumulating', False):
            # .grad already aliased the flat buffer and was not zeroed: it holds REDUCED gradients of earlier steps plus
            # this step's local ones -- a second all-reduce would multiply the earlier part by the world size
            raise RuntimeError('data parallel: gradients were accumulated across steps (no zero_grad between them); '
                               'zero the gradients every step, or all-reduce once after the last accumulation step')
        if self.trace is not None and flat.flat_g.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(flat.flat_g.device))
            self.trace['backward_end'] = ev
        self.grads_ready(flat, 0)          # (the plan joined its side stream before calling: no other producers)
        self._done_upto = None
        self._group_this_backward = None
        if self._fused_opt is not None and getattr(self, '_fused_cover', 0) == flat.total:
            flat.stepped_in_backward = True         # optimizer.step() finds the update done
        self._fused_cover = 0
        for w in self._pending:
            w.wait()
        self._pending = []
        cs = self._stream(flat)
        if cs is not None:
            torch.cuda.current_stream(flat.flat_g.device).wait_stream(cs)
