"""Data-parallel path on CPU: world_size 2, gloo, one process per rank (the N > 1 path of bench.py).

The C ABI is served by the emulator (no GPU here); what is checked is the distributed HOST logic of
segnb.dist: parameter broadcast, the 8-double all-reduce that makes the Jaccard/Dice sums global, the
world-size factor on the backward seed, the bucketed SUM all-reduce of the flat gradient buffer.

Expected values come from the oracle in ONE process: per-shard forward (BatchNorm statistics stay per
rank, as under any torch DP), loss over the concatenated logits (global sums, lib/losses.py:39-42),
(B_total * loss).backward()  ==  what the reference would compute if its batch were split across devices.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'segmentation-networks-benchmark_amd'), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import abi_emulator, train_step_ref
    from segnb import _native as nv
    from segnb import dist as sdist
    from segnb import optim
    nv.set_backend_for_testing(abi_emulator.AbiEmulator())
    sdist.init_from_env(backend='gloo')
    from lib.models.zf_unet import ZF_UNET
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    torch.manual_seed(100 + rank)            # ranks start DIFFERENT: the broadcast must fix that
    m = ZF_UNET(dropout_val=0.0, filters=4).set_compute_dtype('f32').train()
    emu = nv._test_backend
    # SEGNB_DP_RESERVE_CUS = 8: the persistent convolution grids are sized for 96 % = 245 of the 256 emulated CUs (11 left to
    # the collectives); detach() restores the previous value.  Default 0: nothing is touched (unmeasured on hardware: ADVICE r4)
    os.environ['SEGNB_DP_RESERVE_CUS'] = '8'
    dp0 = sdist.DataParallel(m, bucket_bytes=256 << 10)
    assert emu.tuned.get('conv_cu_pct') == 96 and dp0.reserved_cus == 11, (getattr(emu, 'tuned', None), dp0.reserved_cus)
    dp0.detach()
    assert emu.tuned.get('conv_cu_pct') == 100 and dp0.reserved_cus == 0
    del os.environ['SEGNB_DP_RESERVE_CUS']
    emu.tuned.pop('conv_cu_pct')
    dp = sdist.DataParallel(m, bucket_bytes=256 << 10)      # small buckets -> several all-reduces
    assert 'conv_cu_pct' not in emu.tuned and dp.reserved_cus == 0
    x, y = train_step_ref.synthetic_batch(4, 64, seed=77)
    xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
    with torch.no_grad():
        m(xs)                                # builds the flat buffers
    dp.broadcast_parameters(m._engine.flat)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    opt = optim.SGD(m.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = BCEWithLogitsLossAndSmoothJaccard()(m(xs), ys)
    (xs.size(0) * loss).backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    opt.step()
    torch.save({'loss': loss.item(), 'grads': grads, 'sd0': sd0,
                'after': {k: v.clone() for k, v in m.state_dict().items()}},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_matches_oracle(tmp_path):
    from oracle import losses_ref, train_step_ref, zf_unet_ref
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    # broadcast made the replicas identical (rank 0's weights); BN buffers were synchronised too
    for k in r0['sd0']:
        if 'num_batches_tracked' not in k and 'running' not in k:
            assert torch.equal(r0['sd0'][k], r1['sd0'][k]), k
    # both ranks report the GLOBAL loss and hold the same reduced gradients
    assert abs(r0['loss'] - r1['loss']) < 1e-7
    for n in r0['grads']:
        torch.testing.assert_close(r0['grads'][n], r1['grads'][n], rtol=0, atol=0)
    # oracle: same weights, per-shard forward, global loss, B_total seed
    x, y = train_step_ref.synthetic_batch(4, 64, seed=77)
    leaves, work = {}, {}
    for k, v in r0['sd0'].items():
        if zf_unet_ref.is_param(k):
            leaves[k] = v.clone().requires_grad_(True)
    outs = []
    for r in range(2):
        sd = {k: (leaves[k] if k in leaves else v.clone()) for k, v in r0['sd0'].items()}
        outs.append(zf_unet_ref.forward(sd, x[2 * r:2 * r + 2], train=True))
    logits = torch.cat(outs, 0)
    loss = losses_ref.bce_jaccard(logits, y)
    (4 * loss).backward()
    assert abs(loss.item() - r0['loss']) < 1e-5
    for n, p in leaves.items():
        ref = p.grad
        scale = max(float(ref.abs().max()), 1e-6)
        # 2 images x 2x2 pixels per BatchNorm channel at the bottleneck: fp32 summation-order noise is amplified
        # ~1e4x on the way back to the first layer (same effect as in test_plan_cpu.py)
        assert float((r0['grads'][n] - ref).abs().max()) <= 1e-2 * scale + 3e-6, n
    # the fused flat SGD applied the reduced gradient on every rank
    for n, p in leaves.items():
        torch.testing.assert_close(r0['after'][n], r0['sd0'][n] - 1e-3 * r0['grads'][n], rtol=1e-6, atol=1e-7)
        assert torch.equal(r0['after'][n], r1['after'][n])


def _worker_fused(rank, world, port, out_dir, opt_name, late_broadcast=False):
    """The same three steps twice -- optimizer folded into the all-reduce epilogue / ordinary step() -- on two replicas
    of one model inside one job; the emulator's call log tells which launches step() itself made."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'segmentation-networks-benchmark_amd'), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import abi_emulator, train_step_ref
    from segnb import _native as nv
    from segnb import dist as sdist
    from segnb import optim
    emu = abi_emulator.AbiEmulator()
    calls = []
    for name, n_at in (('segnb_sgd_step', 2), ('segnb_rmsprop_step', 3), ('segnb_adam_step', 4)):
        def logged(*a, _f=getattr(emu, name), _n=name, _i=n_at):
            calls.append((_n, int(a[_i])))
            return _f(*a)
        setattr(emu, name, logged)
    nv.set_backend_for_testing(emu)
    sdist.init_from_env(backend='gloo')
    from lib.models.zf_unet import ZF_UNET
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    x, y = train_step_ref.synthetic_batch(4, 64, seed=78)
    xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
    out = {}
    for fused in (True, False):
        torch.manual_seed(5)
        m = ZF_UNET(dropout_val=0.0, filters=4).set_compute_dtype('f32').train()
        dp = sdist.DataParallel(m, bucket_bytes=64 << 10)
        if not late_broadcast:
            with torch.no_grad():
                m(xs)
            dp.broadcast_parameters(m._engine.flat)
        opt = {'sgd': lambda: optim.SGD(m.parameters(), lr=1e-2), 'adam': lambda: optim.Adam(m.parameters(), lr=1e-3),
               'rmsprop': lambda: optim.RMSprop(m.parameters(), lr=1e-3)}[opt_name]()
        if fused:
            dp.fuse_optimizer(opt)
        del calls[:]
        per_step = []
        for it in range(3):
            opt.zero_grad()
            loss = BCEWithLogitsLossAndSmoothJaccard()(m(xs), ys)
            n0 = len(calls)
            (xs.size(0) * loss).backward()
            n1 = len(calls)
            opt.step()
            per_step.append((n1 - n0, len(calls) - n1))
            if late_broadcast and it == 0:
                # bench.py's own order: the first step() builds the flat buffers, THEN the parameters are broadcast (the
                # replicas were seeded alike, so the values do not change) -- ADVICE r2: the first backward used to fuse
                # only the buckets after the first one, stepping those ranges twice (SGD) or raising (Adam)
                dp.broadcast_parameters(m._engine.flat)
        out[fused] = dict(after={k: v.clone() for k, v in m.state_dict().items()}, per_step=per_step,
                          total=m._engine.flat.total, sizes=[c[1] for c in calls],
                          opt_state=opt.state_dict()['state'])
        dp.detach()
    torch.save(out, os.path.join(out_dir, 'fused_rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('opt_name,late_broadcast', [('sgd', False), ('adam', False), ('sgd', True), ('adam', True)])
def test_optimizer_folded_into_allreduce_epilogue(tmp_path, opt_name, late_broadcast):
    """with DataParallel.fuse_optimizer the update of each bucket runs behind its all-reduce, inside
    backward; optimizer.step() launches nothing; parameters, BatchNorm buffers and optimizer state equal the ordinary
    path bit for bit on both ranks.  late_broadcast: the first backward runs BEFORE broadcast_parameters (bench.py's
    order) -- the fusion decision is taken once per backward, at its first bucket."""
    port = _free_port()
    mp.spawn(_worker_fused, args=(2, port, str(tmp_path), opt_name, late_broadcast), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'fused_rank%d.pt' % k), weights_only=False) for k in (0, 1)]
    for k in (0, 1):
        f, u = r[k][True], r[k][False]
        # ordinary path: nothing in backward, one launch in step().  Folded: several bucket launches in backward, none in step()
        assert u['per_step'] == [(0, 1)] * 3
        assert all(b > 1 and s == 0 for b, s in f['per_step']), f['per_step']
        nb = f['per_step'][0][0]
        assert sum(f['sizes'][:nb]) == f['total']                   # the buckets of one step tile the flat buffer
        for name in u['after']:
            assert torch.equal(f['after'][name], u['after'][name]), name
        for idx in u['opt_state']:
            for key, val in u['opt_state'][idx].items():
                assert torch.equal(torch.as_tensor(f['opt_state'][idx][key]), torch.as_tensor(val)), (idx, key)
    for name in r[0][True]['after']:
        if 'running' not in name and 'num_batches' not in name:
            assert torch.equal(r[0][True]['after'][name], r[1][True]['after'][name]), name


def _worker_hipnet(rank, world, port, out_dir, which):
    """An executor-driven model under data parallel: the second backward hands finished gradient ranges to the hook while it
    is still running (Tape.run_closures cuts); the reduced gradients equal the all-reduced gradients of a plain run."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'segmentation-networks-benchmark_amd'), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import warnings
    from oracle import abi_emulator
    from segnb import _native as nv
    from segnb import dist as sdist
    nv.set_backend_for_testing(abi_emulator.AbiEmulator())
    sdist.init_from_env(backend='gloo')
    from lib.losses import BCEWithSigmoidLoss
    from lib.models.tiramisu import FCDenseNet
    from lib.models.unet16 import UNet16

    def make():
        torch.manual_seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if which == 'unet16':
                m = UNet16(num_filters=4)
            elif which == 'linknet34':
                # (its decoder convolutions' biases sit in front of BatchNorms: no closure ever writes their gradients, which used
                # to hold the frontier of finished gradients at the top of the buffer -- no cuts, no early buckets)
                from lib.models.linknet import LinkNet34
                m = LinkNet34()
                for mod in m.modules():
                    if isinstance(mod, torch.nn.Dropout2d):
                        mod.p = 0.0
            else:
                m = FCDenseNet(in_channels=3, down_blocks=(2, 2), up_blocks=(2, 2), bottleneck_layers=2, growth_rate=8,
                               out_chans_first_conv=16, n_classes=1)
                for mod in m.modules():
                    if isinstance(mod, torch.nn.Dropout2d):
                        mod.p = 0.0
        return m.set_compute_dtype('f32').train()
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(2, 3, 32, 32, generator=g)
    y = (torch.rand(2, 1, 32, 32, generator=g) > 0.6).long()
    crit = BCEWithSigmoidLoss()
    # plain run on this rank's shard, gradients summed over the ranks by hand
    ref = make()
    for _ in range(2):
        ref.zero_grad()
        (x.size(0) * crit(ref(x), y)).backward()
    want = {}
    for n, p in ref.named_parameters():
        t = p.grad.clone()
        torch.distributed.all_reduce(t)
        want[n] = t
    # the same under DataParallel: first backward learns the cuts, second uses them
    m = make()
    dp = sdist.DataParallel(m, bucket_bytes=8 << 10)
    with torch.no_grad():
        m(x)
    dp.broadcast_parameters(m._tape.flat)
    launched = []
    orig = dp._launch
    dp._launch = lambda flat, a, b: (launched.append((a, b, len(m._tape.back) if m._tape.back else 0)), orig(flat, a, b))[1]
    calls = []
    ready = dp.grads_ready
    m._grad_ready_hook = lambda flat, lo, producers=(): (calls.append(lo), ready(flat, lo, producers))[1]
    m._grad_ready_hook.__dict__['active'] = True
    per_step = []
    for _ in range(2):
        m.zero_grad()
        n0 = len(calls)
        (x.size(0) * crit(m(x), y)).backward()
        per_step.append(len(calls) - n0)
    got = {n: p.grad.clone() for n, p in m.named_parameters()}
    torch.save(dict(want=want, got=got, per_step=per_step, calls=calls, total=m._tape.flat.total,
                    cuts=dict(m._tape._cuts)), os.path.join(out_dir, 'hipnet_rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('which', ['unet16', 'fcdensenet', 'linknet34'])
def test_executor_models_hand_over_gradients_during_backward(tmp_path, which):
    port = _free_port()
    mp.spawn(_worker_hipnet, args=(2, port, str(tmp_path), which), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'hipnet_rank%d.pt' % k), weights_only=False) for k in (0, 1)]
    for k in (0, 1):
        # first backward: learning pass, no hand-over; second: the hook is called at the cuts, offsets falling
        assert r[k]['per_step'][0] == 0 and r[k]['per_step'][1] >= 1, r[k]['per_step']
        assert all(0 < lo < r[k]['total'] for lo in r[k]['calls']) and r[k]['calls'] == sorted(r[k]['calls'], reverse=True)
        for n in r[k]['want']:
            scale = max(float(r[k]['want'][n].abs().max()), 1e-6)
            assert float((r[k]['got'][n] - r[k]['want'][n]).abs().max()) <= 1e-5 * scale, n
    for n in r[0]['got']:
        assert torch.equal(r[0]['got'][n], r[1]['got'][n]), n


def _worker_wire(rank, world, port, out_dir):
    """One step with the fp32 wire format and one with bf16 buckets, same weights and shard."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'segmentation-networks-benchmark_amd'), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import abi_emulator, train_step_ref
    from segnb import _native as nv
    from segnb import dist as sdist
    nv.set_backend_for_testing(abi_emulator.AbiEmulator())
    sdist.init_from_env(backend='gloo')
    from lib.models.zf_unet import ZF_UNET
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    x, y = train_step_ref.synthetic_batch(4, 64, seed=79)
    xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
    out = {}
    for wire in ('f32', 'bf16'):
        torch.manual_seed(6)
        m = ZF_UNET(dropout_val=0.0, filters=4).set_compute_dtype('f32').train()
        dp = sdist.DataParallel(m, bucket_bytes=64 << 10, wire_dtype=wire)
        m.zero_grad()
        loss = BCEWithLogitsLossAndSmoothJaccard()(m(xs), ys)
        (xs.size(0) * loss).backward()
        out[wire] = {n: p.grad.clone() for n, p in m.named_parameters()}
        dp.detach()
    torch.save(out, os.path.join(out_dir, 'wire_rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_bf16_wire_format_of_the_gradient_buckets(tmp_path):
    """DataParallel(wire_dtype='bf16'): each bucket crosses the links as bf16 and is widened back into the flat fp32
    buffer -- both ranks hold the SAME gradients, equal to the fp32 exchange to bf16 rounding (two roundings: each rank's
    contribution and their sum)."""
    port = _free_port()
    mp.spawn(_worker_wire, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'wire_rank%d.pt' % k), weights_only=False) for k in (0, 1)]
    worst = 0.0
    for n in r[0]['f32']:
        assert torch.equal(r[0]['bf16'][n], r[1]['bf16'][n]), n
        assert torch.equal(r[0]['bf16'][n], r[0]['bf16'][n].to(torch.bfloat16).float()), n     # values ARE bf16 numbers
        ref, got = r[0]['f32'][n], r[0]['bf16'][n]
        scale = max(float(ref.abs().max()), 1e-12)
        worst = max(worst, float((got - ref).abs().max()) / scale)
    assert 0.0 < worst <= 3 * 2.0 ** -8, worst


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment launches the two ranks itself (child
    torch.distributed.run) and forwards rank 0's line: n_gpus 2, ranks_seen 2.  --dry-run stops after the rendezvous
    (gloo here: no GPU in this container; on a GPU box the same path joins over RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['requested_gpus'] == 2, out
