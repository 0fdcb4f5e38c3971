#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: images/sec of the full training step
(torch_train.py:180-190: zero_grad -> model(x) -> loss -> (B*loss).backward() -> optimizer.step()) for
ZF_UNET 224x224, bs=32 per GPU, bf16 compute, BCE+Dice, synthetic tiles resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    N > 1 and no RANK in the environment: bench.py starts its own N ranks (child `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N ... bench.py ...`, before anything touches the GPU) and forwards rank 0's line and
    the job's exit code; under a launcher that has already set RANK / WORLD_SIZE it just joins the job.

Prints ONE JSON line on rank 0.  Besides the throughput it carries
  roofline     : the dominant kernel (the implicit-GEMM convolution) timed live with HIP events on its launch
                 stream during the timed region: achieved = algorithmic FLOPs / event time, against the dense
                 bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline : the oracle's CPU restatement of the same step (oracle/train_step_ref.py, kind "port") on this
                 box's host cores, bounded sample (B=4, a few steps) -- a reported baseline, not a target.
  box          : what THIS box delivers on two fixed probes run before the timed region -- a compute-bound bf16 GEMM
                 (TFLOP/s) and a 1 GB device copy (TB/s).  The boxes of the pool differ by 3-5 % in step time
                 (MI355X_MICROARCH.md, DVFS give-back item 5); `value` / box figures lets two lines from different boxes
                 be compared.
  with_logging_syncs : the same step WITH the reference's per-batch logging (torch_train.py:195-210: loss .item(),
                 gradient abs-max, two metrics, each a host sync) -- SURVEY 8d / BASELINE.md section 2's second number.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'segmentation-networks-benchmark_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

_T_START = time.perf_counter()
import torch
_T_TORCH = time.perf_counter()


class Phases(object):
    """Wall time per phase of this process, printed to stderr as the run proceeds and carried in the line as `phases_s`: where the
    driver's `driver_run_s` goes (BENCH_r05.json: 173 s around a 0.1 s timed region, nothing said where -- VERDICT r5 weak #10)."""

    def __init__(self):
        self.t = _T_TORCH
        self.out = [('import_torch', _T_TORCH - _T_START)]
        print('[bench phase] %-28s %8.2f s' % self.out[0], file=sys.stderr, flush=True)

    def mark(self, name):
        now = time.perf_counter()
        self.out.append((name, now - self.t))
        self.t = now
        print('[bench phase] %-28s %8.2f s' % self.out[-1], file=sys.stderr, flush=True)

    def as_dict(self):
        d = {}
        for k, v in self.out:
            d[k] = round(d.get(k, 0.0) + v, 3)
        d['total'] = round(time.perf_counter() - _T_START, 3)
        return d


PHASES = None

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
GFLOP_PER_IMAGE_224 = 79.26     # conv MACs only, fwd + dgrad + wgrad, no dgrad for the first conv
PEAK_HBM_TBS = 8.0              # HBM3E spec (MI355X_MICROARCH.md; 6.3 TB/s is what a copy achieves)
ALGO_MB_PER_IMAGE_224 = 184.4 + 15.7      # activations 3 (I + O) bf16 per conv + weights / gradients / SGD at bs=32


def cpu_baseline(seconds_budget=20.0):
    """Oracle train step (torch-CPU fp32, all host cores) on B=4 224x224: images/s."""
    from oracle import train_step_ref, zf_unet_ref
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    # thread count from the sweep on the GPU box's 256-core host (tools/cpu_sweep.py, profiles/r05_cpu_sweep.txt): 8 threads 12.0
    # images/s, 16 -> 17.8, 32 -> 12.7, 64 -> 5.2, 128 -> 2.6, 256 -> 0.06 -- oneDNN's B=4 convolutions stop scaling at 16
    torch.set_num_threads(max(1, min(ncpu, 16)))
    B, S = 4, 224
    x, y = train_step_ref.synthetic_batch(B, S, seed=1234)
    sd = zf_unet_ref.default_init_state(filters=32, seed=0)
    train_step_ref.train_step(sd, x, y, 'bce_dice', lr=1e-3)       # warm-up
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < seconds_budget and n < 20):
        train_step_ref.train_step(sd, x, y, 'bce_dice', lr=1e-3)
        n += 1
    dt = time.time() - t0
    return {'value': round(B * n / dt, 3), 'unit': 'images/s', 'cores': torch.get_num_threads(),
            'threads': torch.get_num_threads(), 'host_cores': os.cpu_count(), 'affinity_cores': ncpu, 'kind': 'port',
            'sample': 'oracle/train_step_ref.train_step, ZF_UNET fp32 B=4 224x224 bce_dice SGD, %d steps after '
                      '1 warm-up, torch %s CPU' % (n, torch.__version__)}


def box_calibration(dev):
    """Two fixed probes of the box the line was measured on (NOT part of the product: library GEMM and a device copy):
    a 4096^3 bf16 GEMM repeated for ~50 ms -> TFLOP/s, a 1 GiB device-to-device copy -> TB/s (read + write bytes)."""
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 40
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1)
    tf = 2.0 * 4096 ** 3 * iters / (gemm_ms * 1e-3) / 1e12
    del a, b, c
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.zero_()
    dst.copy_(src)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_ms = e0.elapsed_time(e1) / 5
    tbs = 2.0 * (1 << 30) / (copy_ms * 1e-3) / 1e12
    del src, dst
    torch.cuda.empty_cache()
    return {'gemm_bf16_4096_tflops': round(tf, 1), 'gemm_ms': round(gemm_ms, 1), 'copy_1GiB_TBps': round(tbs, 3),
            'note': 'hipBLASLt bf16 GEMM 4096^3 x %d and a 1 GiB device copy (read + write bytes), before the timed region' % iters}


def kernel_sources_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/*.h): what a committed PMC profile is stamped with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'segmentation-networks-benchmark_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(ROOT, 'segmentation-networks-benchmark_amd', 'csrc', '*.h')) +
                    glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def pmc_profile(model='zf_unet'):
    """The committed PMC traffic profile of this build for `model` (profiles/r*_pmc_traffic[_<model>].json), or None
    when there is none or it was taken from other kernel sources."""
    import glob
    pat = 'r*_pmc_traffic.json' if model == 'zf_unet' else 'r*_pmc_traffic_%s.json' % model
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)))
    if not files:
        return None
    with open(files[-1]) as fh:
        prof = json.load(fh)
    if prof.get('kernel_sources_digest') != kernel_sources_digest():
        return None
    return prof


def pmc_traffic(kernel, model='zf_unet'):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs of this same command, summarised by tools/pmc_traffic.py with the gfx950 corrections of
    MI355X_MICROARCH.md).  Counters cannot be read from inside the run, so the figure is only reported when the
    profile is stamped with the digest of the kernel sources THIS run was built from; otherwise None (a profile of
    other kernels says nothing about this build)."""
    prof = pmc_profile(model)
    if prof is None:
        return None
    k = prof.get('kernels', {}).get(kernel)
    return None if k is None else k['traffic_bytes_per_launch']


def self_launch(n, argv):
    """--gpus N without a launcher: run the N ranks as CHILD processes (never exec: this process may not be replaced
    once anything has touched the GPU, and nothing here has) and return the job's exit code.  Rank 0's JSON line goes
    to the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def rendezvous_check(args):
    """--dry-run: join the job, count the ranks with an all-reduce of ones, print the line's job fields and stop -- the
    launch / rendezvous path of `bench.py --gpus N` without a kernel (gloo on a box without GPUs: tests/test_dist_cpu.py)."""
    import torch.distributed as td
    from segnb import dist as sdist
    sdist.init_from_env()
    ws, rank = sdist.world(), sdist.rank()
    ones = torch.ones(1)
    if ws > 1:
        if torch.cuda.is_available():
            ones = ones.cuda(int(os.environ.get('LOCAL_RANK', '0')))
        td.all_reduce(ones)
    if rank == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': ws, 'ranks_seen': int(ones.item()), 'requested_gpus': args.gpus,
                          'backend': td.get_backend() if ws > 1 else None}))
    return 0 if int(ones.item()) == max(1, args.gpus) else 3


def bench_tiled(args):
    """BASELINE.json configs[4] at the size SURVEY 8d states: inria_submit.predict_tiled (/root/reference/inria_submit.py:237-257)
    over a synthetic 5000 x 5000 x 3 uint8 image -- tile 1024, step 512 (81 tiles), D4 TTA (648 forward items), batch 4,
    pyramid weights -- on segnb.tiled.predict_tiled with the default UNet16 (bf16, eval).  One "step" = one image.  Reports
    tiles/s, ms per image, the GPU-time share of upload / gather / forward / logits copy / merge / download (HIP events between
    the phases) and the fraction of the model's own eval-forward rate at batch 4 measured in the same process."""
    import numpy as np
    from segnb import dist as sdist
    from segnb import _native as nv
    from segnb.engine import InputNorm
    from segnb.tiled import predict_tiled
    import warnings
    sdist.init_from_env()
    ws, rank = sdist.world(), sdist.rank()
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) if ws > 1 else 0)
    torch.cuda.set_device(dev)
    nv.load()
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from lib.models.unet16 import UNet16
        model = UNet16().set_compute_dtype(args.dtype).to(dev).eval()
    S, B = args.size or 1024, args.batch or 4
    side = args.image_size
    rng = np.random.RandomState(1234)
    image = rng.randint(0, 256, size=(side, side, 3), dtype=np.uint8)
    norm = InputNorm(mean=(0.40, 0.42, 0.38), std=(0.19, 0.18, 0.18))       # (INRIA_MEAN / INRIA_STD stand-ins: lib/datasets/Inria.py:34)
    # the model's own eval-forward rate at this batch (the ceiling predict_tiled is measured against)
    x = torch.randn(B, 3, S, S, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nf = 30
        for _ in range(nf):
            model(x)
        torch.cuda.synchronize()
        fwd_ips = B * nf / (time.perf_counter() - t0)
    for _ in range(max(1, args.warmup)):
        predict_tiled(image, model, norm, S, B)
    if ws > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    timing, t0 = {}, time.perf_counter()
    for i in range(args.steps):
        mask = predict_tiled(image, model, norm, S, B, timing=timing if i == args.steps - 1 else None)
    torch.cuda.synchronize()
    if ws > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if ws > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    items = timing['nitems']
    ms_img = dt / args.steps * 1e3
    tiles_s = items * args.steps / dt
    tot = sum(timing['ms'].values())
    line = {
        'metric': 'tile forwards/sec: predict_tiled (UNet16 %dx%d tiles, step %d, D4 TTA x8) over a %dx%dx3 uint8 image'
                  % (S, S, S // 2, side, side),
        'value': round(tiles_s, 2), 'unit': 'tiles/s', 'n_gpus': ws, 'steps': args.steps, 'warmup': max(1, args.warmup),
        'ms_per_step': round(ms_img, 2), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[4]: TernausNet VGG16-UNet, lib/tiles.py sliding-window inference '
                               '(inria_submit.predict_tiled), one step = one image', 'image': '%dx%dx3 uint8' % (side, side),
                   'tile': S, 'tile_step': S // 2, 'tiles': timing['ntiles'], 'items_per_image': items, 'batch': B,
                   'forward_batches_per_image': timing['batches'], 'weight': 'pyramid'},
        'tiled': {'ms_per_image': round(ms_img, 2),
                  'gpu_ms_by_phase_last_image': {k: round(v, 2) for k, v in timing['ms'].items()},
                  'share_by_phase': {k: round(v / tot, 4) for k, v in timing['ms'].items()},
                  'eval_forward_images_per_s_batch%d' % B: round(fwd_ips, 1),
                  'fraction_of_eval_forward_ceiling': round(tiles_s / (fwd_ips * ws), 4),
                  'note': ('gather of batch k+1 and the logits copy of batch k run on the launch stream between the forwards: '
                          'together < 2 %% of the image (share_by_phase), so no side stream is used; the merge is one launch '
                          'at the end over the logits of all %d items kept in HBM') % items,
                  'mask_mean': round(float(mask.mean()), 6)},
    }
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the model\'s configuration)')
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--loss', default='bce_dice', choices=['bce_dice', 'bce_jaccard', 'bce'])
    ap.add_argument('--model', default='zf_unet', choices=['zf_unet', 'linknet34', 'fcdensenet103', 'fcdensenet67', 'unet16'],
                    help='zf_unet = the headline metric (BASELINE.json); the others time the remaining SURVEY 8d rows at '
                         'their own sizes: linknet34 512x512 bs=16, fcdensenet103 256x256 bs=8, unet16 1024x1024 bs=4')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-box', action='store_true',
                    help='skip the box calibration probes (profiling runs: their GEMM / copy kernels would be counted)')
    ap.add_argument('--fuse-optimizer', action='store_true',
                    help='N > 1: SGD update of each gradient bucket behind its all-reduce (segnb.dist.DataParallel.fuse_optimizer)')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--timed-only', action='store_true',
                    help='stop the GPU work after the timed region: no host-enqueue probe, no logged steps, no kernel timer, no CPU '
                         'baseline, no box probes (for kernel traces: every step of the run is then a warm-up or a timed step)')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the whole training step from one captured HIP graph (auto = off: eager launches overlap the weight-gradient stream better)')
    ap.add_argument('--wire', default='f32', choices=['f32', 'bf16'],
                    help='N > 1: wire format of the gradient all-reduce buckets (bf16 halves the bytes on xGMI; the sum is '
                         'taken in bf16 by RCCL -- off by default: the headline keeps the fp32 exchange)')
    ap.add_argument('--tiled', action='store_true',
                    help='BASELINE.json configs[4] as a bench line: predict_tiled over a synthetic 5000x5000x3 uint8 image with '
                         'UNet16 (tile 1024 / step 512 / D4 TTA / batch 4); one step = one image (default --steps 3 --warmup 1)')
    ap.add_argument('--image-size', type=int, default=5000, help='--tiled: side of the synthetic image')
    ap.add_argument('--dry-run', action='store_true',
                    help='join the job, count the ranks (all-reduce of ones), print n_gpus / ranks_seen and stop')
    args = ap.parse_args()
    if args.timed_only:
        args.no_cpu_baseline = args.no_kernel_timer = args.no_box = True

    if args.gpus > 1 and 'RANK' not in os.environ:
        # nothing above has touched the GPU (importing torch and parsing arguments do not)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.dry_run:
        sys.exit(rendezvous_check(args))
    if args.tiled:
        if '--steps' not in sys.argv:
            args.steps = 3
        if '--warmup' not in sys.argv:
            args.warmup = 1
        return bench_tiled(args)

    global PHASES
    PHASES = ph = Phases()
    from segnb import dist as sdist
    from segnb import engine, optim
    from segnb import _native as nv
    from lib.models.zf_unet import ZF_UNET
    from lib import losses as L
    ph.mark('import_package')

    sdist.init_from_env()
    ws, rank = sdist.world(), sdist.rank()
    if ws != max(1, args.gpus) and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, ws), file=sys.stderr)
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) if ws > 1 else 0)
    torch.cuda.set_device(dev)
    nv.load()
    torch.cuda.synchronize()
    ph.mark('library_load_and_gpu_init')
    box = box_calibration(dev) if (rank == 0 and not args.no_box) else None
    ph.mark('box_probes')

    # (model constructor, images per GPU, size, algorithmic GFLOP per image fwd+bwd at that size -- SURVEY 8d)
    import warnings
    models = {'zf_unet': (ZF_UNET, 32, 224, GFLOP_PER_IMAGE_224),
              'linknet34': (lambda: __import__('lib.models.linknet', fromlist=['x']).LinkNet34(), 16, 512, 138.5),
              'fcdensenet103': (lambda: __import__('lib.models.tiramisu', fromlist=['x']).FCDenseNet103(n_classes=1), 8, 256, 156.1),
              'fcdensenet67': (lambda: __import__('lib.models.tiramisu', fromlist=['x']).FCDenseNet67(n_classes=1), 8, 256, 6 * 27.11),
              'unet16': (lambda: __import__('lib.models.unet16', fromlist=['x']).UNet16(), 4, 1024, 3 * 1278.8)}
    ctor, dB, dS, gflop_at = models[args.model]
    args.batch = args.batch or dB
    args.size = args.size or dS
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ctor().set_compute_dtype(args.dtype).to(dev).train()
    if args.model != 'zf_unet':
        args.no_cpu_baseline = True          # the CPU baseline is the headline step (oracle ZF_UNET)
    crit = {'bce_dice': L.BCEAndDiceLoss, 'bce_jaccard': L.BCEWithLogitsLossAndSmoothJaccard,
            'bce': L.BCEWithSigmoidLoss}[args.loss]()
    opt = optim.SGD(model.parameters(), lr=1e-3)
    dp = sdist.DataParallel(model, wire_dtype=args.wire)
    if args.fuse_optimizer:
        dp.fuse_optimizer(opt)
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, 3, S, S, generator=g).to(dev)
    y = (torch.rand(B, 1, S, S, generator=g) > 0.7).long().to(dev)

    def flat_of():
        return model._engine.flat if args.model == 'zf_unet' else model._tape.flat

    def step_keep_grads():
        # zero_grad(set_to_none=False) semantics without touching .grad objects: the backward plan clears the
        # flat gradient buffer itself when every .grad already aliases it
        flat_of().flat_g.zero_()
        out = model(x)
        loss = crit(out, y)
        (x.size(0) * loss).backward()
        opt.step()
        return loss

    def step():
        opt.zero_grad()
        out = model(x)
        loss = crit(out, y)
        (x.size(0) * loss).backward()
        opt.step()
        return loss

    # (the dependent chain on a high-priority stream with the weight gradients on a normal-priority one was measured in rounds 2
    # and 4: the same step time either way, the dispatcher does not prefer the chain)
    ph.mark('model_and_inputs')
    loss = step()                          # builds the plan, flat buffers
    torch.cuda.synchronize()
    ph.mark('first_step_plan_recording')
    dp.broadcast_parameters(flat_of())
    for _ in range(max(0, args.warmup - 1)):
        loss = step()
    torch.cuda.synchronize()
    ph.mark('warmup_steps')

    # ---- whole-step HIP graph: ~300 kernel launches per step are replayed from ONE graph launch ----------------
    # 'auto' = eager launches: the weight gradients run on a second stream beside the data-gradient chain
    # (segnb.engine.Runtime.fork_side), which the eager path overlaps well (6.95 ms/step) while the same step replayed
    # from one HIP graph measured 7.5 ms (7.4 without the second stream); the host needs 4.6 ms to enqueue a step
    use_graph = args.graph == 'on'
    graph = None
    if use_graph:
        opt.zero_grad(set_to_none=False)          # keep parameter.grad as views of the flat gradient buffer
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step_keep_grads()
        torch.cuda.synchronize()

    def run_step():
        if graph is not None:
            graph.replay()
            return static_loss
        return step()

    if ws > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
    torch.cuda.synchronize()
    if ws > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    ph.mark('timed_region')
    ranks_seen = 1
    if ws > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)            # every rank that timed the region adds one
        ranks_seen = int(ones.item())

    # host time to ENQUEUE one step (no synchronisation inside; the GPU is still busy with the first step when the last
    # one has been issued unless the host is the slower side) -- reported beside the step time, outside the timed region
    host_ms = None
    if graph is None and not args.timed_only:
        # three steps from an idle GPU, best of three: a longer unsynchronised run can fill the HIP queue, and the host
        # then waits for the GPU inside a launch call (seen as 2.1-2.3 ms "enqueue" time on some boxes)
        for _ in range(3):
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(3):
                loss_h = step()
            t = (time.perf_counter() - th) / 3 * 1e3
            host_ms = t if host_ms is None else min(host_ms, t)
        torch.cuda.synchronize()
    ph.mark('host_enqueue_probe')

    # ---- the same step WITH the reference's per-batch logging (torch_train.py:195-210): loss .item(), the global gradient
    # abs-max ('train/grad/global_abs_max': one fused reduction over the flat gradient buffer here, a per-parameter loop of
    # .abs().max().cpu().item() there), and the two metrics -- every one a host sync.  SURVEY 8d / BASELINE.md section 2.
    logging_ms = None
    if graph is None and ws == 1 and not args.timed_only:
        import torch_train as tt
        mets = tt.default_metrics()

        def step_logged():
            opt.zero_grad()
            out = model(x)
            loss_l = crit(out, y)
            (x.size(0) * loss_l).backward()
            opt.step()
            vals = [loss_l.cpu().item(), tt.grad_global_abs_max(model)]
            for m in mets.values():
                vals.append(m(out, y).cpu().item())
            return vals

        step_logged()
        torch.cuda.synchronize()
        tl = time.perf_counter()
        nlog = min(args.steps, 10)
        for _ in range(nlog):
            logged = step_logged()
        torch.cuda.synchronize()
        logging_ms = (time.perf_counter() - tl) / nlog * 1e3
    ph.mark('logging_steps')

    # ---- live per-kernel timing (HIP events on the launch stream).  Event records cannot sit inside a replayed
    # graph, so when the timed region ran from the graph the same step is run eagerly right after it, with the
    # events around every convolution launch; without a graph the events are recorded in the timed region itself.
    timer = None
    if not args.no_kernel_timer:
        timer = engine.KernelTimer()
        engine.TIMER = timer
        timer_steps = min(args.steps, 5)
        # ZF_UNET: the first of these steps records launch lists that contain the timing events, the others replay them
        # -- the kernels are timed under the launcher of the timed region; the executor models launch eagerly here
        # (no synchronisation between them: the last step runs in the steady state of the timed region, the GPU never
        # waiting for the host; replayed events hold the records of that last step, eager ones those of all steps)
        for _ in range(timer_steps):
            loss_t = step()
        torch.cuda.synchronize()
        if timer.persistent:
            timer_steps = 1
        timer.collect()
        engine.TIMER = None
    ph.mark('kernel_timer_steps')
    final_loss = float(loss.item())

    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    value = ws * B * args.steps / dt
    gflop_img = gflop_at * (S / float(dS)) ** 2
    peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else PEAK_F32_TFLOPS
    out = {
        'metric': ('images/sec/GPU (fwd+bwd) ZF_UNET 224x224 bs=32; 1/2/4/8-GPU scaling' if args.model == 'zf_unet' else
                   'images/sec/GPU (fwd+bwd) %s %dx%d bs=%d (SURVEY 8d row, not the headline metric)' % (args.model, S, S, B)),
        'value': round(value, 2), 'unit': 'images/s', 'per_gpu': round(value / ws, 2),
        'n_gpus': ws, 'ranks_seen': ranks_seen, 'grad_wire': args.wire if ws > 1 else None, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': ('ZF_UNET %dx%d %s bs=%d/GPU, %s, SGD lr 1e-3, Dropout2d 0.2, train step '
                                'torch_train.py:180-190 (configs[1])' % (S, S, args.dtype, B, args.loss))
                               if args.model == 'zf_unet' else
                               '%s %dx%d %s bs=%d/GPU, %s, SGD lr 1e-3, train step torch_train.py:180-190'
                               % (args.model, S, S, args.dtype, B, args.loss),
                   'global_batch': B * ws, 'parallelism': 'dp%d' % ws},
        'final_loss': round(final_loss, 6), 'hip_graph': bool(graph is not None),
        'optimizer_in_allreduce_epilogue': bool(args.fuse_optimizer and dp.active),
        # forward / backward launch lists recorded once and replayed from C (segnb_plan_run) in the timed region
        'launch_plan': bool(any(p[0] for p in model._engine._cplans.values()) if args.model == 'zf_unet' else
                            any(e.get('state') == 'ready' for e in model._tape.plans.values())),
        'host_enqueue_ms_per_step': None if host_ms is None else round(host_ms, 3),
        'step_mfma_frac': round(value / ws * gflop_img / 1e3 / peak, 4),
        'box': box,
        'with_logging_syncs': None if logging_ms is None else {
            'ms_per_step': round(logging_ms, 3), 'value': round(B / (logging_ms * 1e-3), 2), 'unit': 'images/s',
            'what': 'the timed step + loss.item() + grad_global_abs_max + %d metrics, each a host sync '
                    '(torch_train.py:195-210)' % len(mets)},
    }
    if timer is not None:
        summ = timer.summary()
        # dominant family = the one that carries most of the step's algorithmic FLOPs (forward + data gradient: 2/3).
        # By stretched in-situ time the two families tie and the choice would flip from run to run: the weight
        # gradients run on a second stream BESIDE the dependent chain, so their launch durations include the overlap.
        dom = max(summ, key=lambda k: summ[k][2])
        n, tot_ms, tot_fl = summ[dom]
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        fams = {'conv_fprop': 'segnb_conv_fprop launches: conv_fprop_ws_kernel / conv_roll_kernel / conv_fprop_rw_kernel / '
                              'conv_fprop_c8_kernel / conv_fprop_s1x9_kernel / conv_fprop_kernel (forward + data gradient)',
                'conv_wgrad': 'segnb_conv_wgrad launches: conv_wgrad_s1x9_kernel / conv_wgrad_kernel'}
        out['kernel_sources_digest'] = kernel_sources_digest()
        # `achieved` / `frac` count the FLOPs the launches EXECUTE; the ALGORITHMIC rate of SURVEY 8d (the reference's 3x3 taps
        # over every input channel) is reported beside it -- where a launch issues fewer (the low-resolution data gradient
        # of an upsampled segment: 16 instead of 36 multiply-adds per low-resolution pixel) the algorithmic figure is higher
        ex_ratio = (timer.executed.get(dom, 0.0) / timer.algorithmic[dom]) if timer.algorithmic.get(dom) else 1.0
        out['roofline'] = {'kernel': fams.get(dom, dom), 'bound': 'mfma', 'achieved': round(ach * ex_ratio, 2), 'peak': peak,
                           'unit': 'TFLOP/s', 'frac': round(ach * ex_ratio / peak, 4),
                           'algorithmic_tflops': round(ach, 2), 'algorithmic_frac': round(ach / peak, 4),
                           'traffic': pmc_traffic(dom, args.model),
                           'launches_per_step': n // timer_steps,
                           'avg_launch_us': round(tot_ms / n * 1e3, 2),
                           'flops_per_launch': round(tot_fl / n),
                           'share_of_step': round((tot_ms / timer_steps) / (dt * 1e3 / args.steps), 4)}
        out['kernels'] = {k: {'launches_per_step': v[0] // timer_steps, 'ms_per_step': round(v[1] / timer_steps, 3),
                              'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in summ.items()}
    # second roofline figure: HBM.  bytes_per_step = measured HBM traffic of one step (PMC passes of this same command,
    # stamped with the kernel digest: tools/pmc_traffic.py); achieved = those bytes / the step time of THIS run.
    prof = pmc_profile(args.model)
    step_bytes = prof.get('bytes_per_step') if prof else None
    algo_bytes = ALGO_MB_PER_IMAGE_224 * 1e6 * B * (S / 224.0) ** 2 if args.model == 'zf_unet' else None
    hbm = {'peak': PEAK_HBM_TBS, 'unit': 'TB/s', 'bytes_per_step': step_bytes,
           'algorithmic_bytes_per_step': None if algo_bytes is None else round(algo_bytes)}
    if step_bytes:
        hbm['achieved'] = round(step_bytes / (ms * 1e-3) / 1e12, 3)
        hbm['frac'] = round(hbm['achieved'] / PEAK_HBM_TBS, 4)
        if algo_bytes:
            hbm['traffic_over_algorithmic'] = round(step_bytes / algo_bytes, 3)
    out['roofline_hbm'] = hbm
    if hbm.get('frac') is not None:
        # the step sits under BOTH roofs at once (`roofline.frac` is a fraction of the MFMA peak, `roofline_hbm.frac` of the
        # HBM peak); this names the one the whole step is closer to
        out['step_bound'] = 'hbm' if hbm['frac'] > out['step_mfma_frac'] else 'mfma'
    if ws == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
        ph.mark('cpu_baseline')
    out['phases_s'] = ph.as_dict()
    print(json.dumps(out))


if __name__ == '__main__':
    try:
        main()
    finally:
        import torch.distributed as _td
        if _td.is_available() and _td.is_initialized():
            _td.destroy_process_group()
