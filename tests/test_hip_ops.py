"""-m gpu: every HIP kernel behind the C ABI vs (a) the ABI emulator (oracle/abi_emulator.py) run on CPU
copies of the same inputs and (b) where one exists, the torch-CPU fp32 operator the reference would have
called.  Calls go through segnb._native (ctypes -> libsegnb_hip.so): no torch operator computes anything
on the GPU side.

Tolerances (written per check):
  f32  path : exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) vs CPU fp32 -> summation-order noise, rtol 1e-4 on
              the tensor scale
  bf16 path : identical bf16 storage rounding on both sides, fp32 accumulation -> at most 1-2 bf16 ulps
              (2^-8 relative) on individual elements after re-rounding
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import abi_emulator
from segnb import _native as nv
from segnb import convplan as cp
from segnb.engine import ConvOp, PackTable, Runtime, UpCatConvOp, View

pytestmark = pytest.mark.gpu

EMU = abi_emulator.AbiEmulator()
DTYPES = ['f32', 'bf16']


def tol(dtype):
    return dict(rtol=2e-2, atol=2e-2) if dtype == 'bf16' else dict(rtol=2e-4, atol=2e-4)


def check(name, got, ref, dtype, scale=None):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    s = max(s, 1e-6)
    t = tol(dtype)
    err = (got - ref).abs()
    bound = t['atol'] * s + t['rtol'] * ref.abs()
    bad = err > bound
    assert not bool(bad.any()), '%s [%s]: %d/%d elements off, max err %.3g (scale %.3g), first bad idx %s' % (
        name, dtype, int(bad.sum()), bad.numel(), float(err.max()), s, bad.nonzero()[:4].tolist())


class on_emulator(object):
    def __enter__(self):
        nv.set_backend_for_testing(EMU)

    def __exit__(self, *a):
        nv.set_backend_for_testing(None)


def torch_bn_act_bwd(y, g, coef, act, slope, dtype, res=None):
    """Oracle of the BatchNorm-backward reduction pass (lib/modules/abn/functions.py:107-112: dz through the activation, then
    edz / eydz) with torch's own autograd on the CPU: z = (y - mean) * scale + shift (+ residual), a = act(z), dz = d a / d z * g
    rounded to the activation dtype, sums in float64.  y, g, res: [N, H, W, C] tensors of the activation dtype; coef [4][C] =
    scale, shift, mean, invstd.  -> (dz, sum dz, sum dz * yhat)"""
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    co = coef.detach().float().cpu()
    yf = y.detach().float().cpu()
    z = ((yf - co[2]) * co[0] + co[1]).requires_grad_(True)
    zz = z if res is None else z + res.detach().float().cpu()
    a = {nv.ACT_RELU: torch.relu, nv.ACT_LEAKY: lambda t: F.leaky_relu(t, slope), nv.ACT_NONE: lambda t: t}[act](zz)
    a.backward(g.detach().float().cpu())
    dz = z.grad.to(tdt)
    dd = dz.double().reshape(-1, dz.shape[-1])
    yh = ((yf - co[2]) * co[3]).double().reshape(-1, dz.shape[-1])
    return dz, dd.sum(0), (dd * yh).sum(0)


def test_library_loads_and_reports_device():
    lib = nv.load()
    assert lib.segnb_version() >= 1
    assert lib.segnb_device_cus() > 0


def test_bad_arguments_raise_runtime_error():
    g = nv.ConvGeom()           # all zeros -> rejected by check_geom, status -> RuntimeError
    with pytest.raises(RuntimeError):
        nv.call('segnb_conv_fprop', g, nv.F32, 1, 1, None, 0, 1, None, 0)


# ------------------------------------------------------------------------------------------------------
# convolutions: forward, data gradient, weight gradient, for every flavour the four models use
# ------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # name,              N, H,  W,  segs(real,pad),        Co, k, s, p, transposed
    ('3x3 first layer',  2, 20, 28, [(3, 8)],              32, 3, 1, 1, False),
    ('3x3 32->32',       3, 17, 19, [(32, 32)],            32, 3, 1, 1, False),
    ('3x3 concat pad',   2, 12, 12, [(12, 16), (6, 8)],    6,  3, 1, 1, False),
    ('3x3 wide',         2, 14, 14, [(256, 256)],          192, 3, 1, 1, False),
    ('3x3 deep 7x7',     4, 7,  7,  [(512, 512)],          256, 3, 1, 1, False),
    ('1x1',              2, 9,  11, [(64, 64)],            16, 1, 1, 0, False),
    ('7x7 s2 stem',      2, 32, 32, [(3, 8)],              64, 7, 2, 3, False),
    ('3x3 s2',           2, 16, 18, [(64, 64)],            128, 3, 2, 1, False),
    ('1x1 s2',           2, 16, 16, [(64, 64)],            128, 1, 2, 0, False),
    ('3x3 p0',           1, 13, 13, [(32, 32)],            32, 3, 1, 0, False),
    ('2x2 p1',           1, 11, 11, [(32, 32)],            8,  2, 1, 1, False),
    # the rolling-window kernel (fprop_roll.hip): valid 3 x 3 convolution (linknet.py:60: Conv2d(32, 32, 3)) and its data gradient
    # (input and output grids differ), 32 -> 64 / 64 -> 32 with the channels split over two waves of a strip
    ('roll 3x3 p0',      2, 41, 45, [(32, 32)],            32, 3, 1, 0, False),
    ('roll 3x3 p0 co24', 1, 36, 70, [(32, 32)],            24, 3, 1, 0, False),
    ('roll 32->64',      2, 33, 47, [(32, 32)],            64, 3, 1, 1, False),
    ('roll 64->32',      2, 33, 47, [(64, 64)],            32, 3, 1, 1, False),
    # the first layer's rolling weight gradient (wgrad_roll.hip: conv_wgrad_c8roll_kernel, 8 padded input channels, rows of >= 32
    # pixels): ragged strips and row segments, 24 and 16 output channels (one or two 16-channel halves per wave)
    ('c8 roll ragged',   2, 37, 75, [(3, 8)],              24, 3, 1, 1, False),
    ('c8 roll co16',     3, 9,  40, [(5, 8)],              16, 3, 1, 1, False),
    # stride-1 3x3 shapes that take the transposing-LDS-read weight-gradient kernel (wgrad_s1.hip)
    ('3x3 s1x9 thin',    2, 21, 37, [(32, 32)],            32, 3, 1, 1, False),
    ('3x3 s1x9 thin cat', 1, 16, 56, [(64, 64), (30, 32)], 24, 3, 1, 1, False),
    ('3x3 s1x9 thin p0', 1, 30, 30, [(16, 16)],            32, 3, 1, 0, False),
    ('3x3 s1x9 64x64',   2, 10, 36, [(80, 80)],            72, 3, 1, 1, False),
    ('3x3 s1x9 64x64 b', 1, 28, 28, [(128, 128)],          128, 3, 1, 1, False),
    ('3x3 s1x9 w16',     3, 14, 14, [(96, 96)],            160, 3, 1, 1, False),
    ('3x3 s1x9 co96',    1, 20, 40, [(32, 32)],            96, 3, 1, 1, False),      # 96-channel output tiles
    ('3x3 s1x9 co192',   1, 18, 36, [(64, 64)],            192, 3, 1, 1, False),
    ('3x3 s1x9 flat7',   6, 7, 7,   [(96, 96)],            72, 3, 1, 1, False),      # 4 images / iteration, ragged N
    ('3x3 s1x9 flat7 cat', 5, 7, 7, [(64, 64), (40, 40)],  136, 3, 1, 1, False),
    ('3x3 s1x9 flat14',  2, 14, 14, [(64, 64)],            64, 3, 1, 1, False),
    # few pixels x few output channels x deep K: the register-direct split-K forward (conv_fprop_deepk_kernel)
    ('3x3 deep K',       8, 16, 16, [(1040, 1040)],        16, 3, 1, 1, False),
    ('3x3 deep K cat',   3, 9,  12, [(512, 512), (520, 528)], 16, 3, 1, 1, False),
    ('1x1 deep K',       2, 8,  8,  [(2304, 2304)],        24, 1, 1, 0, False),
    # at most 8 output channels, >= 1024 output pixels: the FMA weight-gradient kernel (conv_wgrad_co8_kernel)
    ('2x2 head co1',     2, 40, 48, [(32, 32)],            1,  2, 1, 1, False),
    ('3x3 co6 cat',      1, 36, 40, [(12, 16), (6, 8)],    6,  3, 1, 1, False),
    ('3x3 s2 co8',       2, 48, 40, [(16, 16)],            8,  3, 2, 1, False),
    # stride-2 windows wide / tall enough for the tile kernel of the strided weight gradients (conv_wgrad_sx_kernel): LinkNet34's
    # stem (linknet.py:16), the 3x3 stride-2 convolutions of ResNet34's layer2..4, the transposed 3x3 stride-2 finaldeconv1
    # (linknet.py:41: its weight gradient gathers dy with stride 2)
    ('7x7 s2 stem tile', 2, 64, 72, [(3, 8)],              64, 7, 2, 3, False),
    ('7x7 s2 stem ragged', 1, 70, 90, [(3, 8)],            40, 7, 2, 3, False),
    ('3x3 s2 tile',      2, 50, 54, [(64, 64)],            128, 3, 2, 1, False),
    ('3x3 s2 tile w16',  2, 28, 30, [(96, 96)],            72, 3, 2, 1, False),
    ('convT 3x3 s2 tile', 2, 20, 26, [(64, 64)],           32, 3, 2, 0, True),
    ('convT 3x3 s2 tile w16', 1, 14, 16, [(40, 40)],       64, 3, 2, 0, True),
    # 1x1 stride 1 (LinkNet34's decoder blocks, linknet.py:27-35; FCDenseNet's transition-down, tiramisu.py:49): the batch as one
    # row of 256-pixel tiles, 16 / 32 / 64 / 128 input channels per block
    ('1x1 tile c16',     2, 20, 24, [(16, 16)],            64, 1, 1, 0, False),
    ('1x1 tile c32',     1, 16, 17, [(32, 32)],            24, 1, 1, 0, False),
    ('1x1 tile c40',     3, 16, 16, [(40, 40)],            72, 1, 1, 0, False),
    ('1x1 tile c112',    1, 18, 20, [(112, 112)],          112, 1, 1, 0, False),
    ('1x1 tile c304',    2, 16, 16, [(304, 304)],          304, 1, 1, 0, False),
    ('convT 4x4 s2 tile', 2, 24, 26, [(72, 72)],           64, 4, 2, 1, True),       # (unet16.py:30)
    ('convT 4x4 s2 tile w16', 1, 14, 12, [(64, 64)],       32, 4, 2, 1, True),
    # ConvTranspose2d(4, 2, 1) forward as ONE launch of the four phases (segnb_upconv_fprop): >= 128 input channels
    ('convT 4x4 s2 upf', 2, 16, 20, [(128, 128)],          64, 4, 2, 1, True),
    ('convT 4x4 s2 upf co32', 1, 24, 24, [(192, 192)],     32, 4, 2, 1, True),
    ('convT 4x4 s2 upf co24', 1, 14, 12, [(128, 128)],     24, 4, 2, 1, True),
    ('2x2 tile co24',    1, 30, 33, [(64, 64)],            24, 2, 1, 1, False),      # (and '2x2 head co1' above: linknet.py:45)
    ('convT 4x4 s2 p1',  2, 8,  9,  [(32, 32)],            32, 4, 2, 1, True),
    ('convT 3x3 s2 p0',  2, 7,  8,  [(48, 48)],            40, 3, 2, 0, True),
]


def _run_conv(device, dtype, case, w, b, x_nchw, dy_nchw):
    name, N, H, W, segs, Co, k, s, p, transposed = case
    rt = Runtime(device, dtype)
    wt = w.to(device)
    bt = b.to(device)
    op = ConvOp(rt, wt, bt, segs, s, p, transposed, need_dgrad=True)
    op.pack(H, W)
    Ho, Wo = op.out_hw(H, W)
    # input view = channel slices of a wider buffer (exercises ld != C and padded concat segments)
    ld_in = op.Cip + 8
    xbuf = rt.zeros((N, H, W, ld_in))
    xv = View(xbuf, N, H, W, op.Cip, ld_in, 8)
    off, roff = 0, 0
    dense = xv.dense()
    for real, padded in segs:
        dense[..., off:off + real] = x_nchw[:, roff:roff + real].permute(0, 2, 3, 1).to(device, rt.tdtype)
        off += padded
        roff += real
    yv = View.alloc(rt, N, Ho, Wo, op.Cop)
    stats = rt.zeros((16, 2, op.Cop), torch.float64)       # SEGNB_STAT_REPLICAS copies
    op.fprop(xv, yv, stats)
    dyv = View.alloc(rt, N, Ho, Wo, op.Cop)
    dyv.dense()[..., :Co] = dy_nchw.permute(0, 2, 3, 1).to(device, rt.tdtype)
    dxv = View.alloc(rt, N, H, W, op.Cip)
    op.dgrad(dyv, dxv)
    gw = torch.zeros_like(wt)
    op.wgrad(xv, dyv, gw)
    if device != 'cpu':
        torch.cuda.synchronize()
    return (yv.dense().float().cpu(), stats.sum(0).cpu(), dxv.dense().float().cpu(), gw.cpu(), segs, Co)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fprop_dgrad_wgrad(case, dtype):
    name, N, H, W, segs, Co, k, s, p, transposed = case
    Ci = sum(r for r, _ in segs)
    gen = torch.Generator().manual_seed(hash(name) % 1000)
    wshape = (Ci, Co, k, k) if transposed else (Co, Ci, k, k)
    w = torch.randn(wshape, generator=gen) * (2.0 / (Ci * k * k)) ** 0.5
    b = torch.randn(Co, generator=gen) * 0.1
    x = torch.randn(N, Ci, H, W, generator=gen)
    if transposed:
        Ho, Wo = cp.convt_out_size(H, k, s, p), cp.convt_out_size(W, k, s, p)
    else:
        Ho, Wo = cp.conv_out_size(H, k, s, p), cp.conv_out_size(W, k, s, p)
    dy = torch.randn(N, Co, Ho, Wo, generator=gen)
    if dtype == 'bf16':      # identical rounded operands on both sides and in the torch reference
        w, x, dy = (t.bfloat16().float() for t in (w, x, dy))
    y_g, st_g, dx_g, gw_g, _, _ = _run_conv('cuda', dtype, case, w, b, x, dy)
    with on_emulator():
        y_e, st_e, dx_e, gw_e, _, _ = _run_conv('cpu', dtype, case, w, b, x, dy)
    # (a) HIP vs emulator, including pad channels (must be exactly zero on both)
    check(name + ' y', y_g, y_e, dtype)
    check(name + ' dx', dx_g, dx_e, dtype)
    check(name + ' dW', gw_g, gw_e, dtype if dtype == 'f32' else 'f32', scale=float(gw_e.abs().max()) * (20 if dtype == 'bf16' else 1))
    np.testing.assert_allclose(st_g.numpy(), st_e.numpy(), rtol=2e-3 if dtype == 'bf16' else 1e-5,
                               atol=(2e-2 if dtype == 'bf16' else 1e-4) * float(st_e.abs().max()))
    assert float(y_g[..., Co:].abs().max()) == 0.0 if y_g.shape[-1] > Co else True
    # (b) HIP vs the torch operator the reference calls
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    if transposed:
        yr = F.conv_transpose2d(xr, wr, b, stride=s, padding=p)
    else:
        yr = F.conv2d(xr, wr, b, stride=s, padding=p)
    yr.backward(dy)
    check(name + ' y vs torch', y_g[..., :Co].permute(0, 3, 1, 2), yr, dtype)
    # dx: gather the real channels out of the padded segments
    parts, off = [], 0
    for real, padded in segs:
        parts.append(dx_g[..., off:off + real])
        off += padded
    check(name + ' dx vs torch', torch.cat(parts, -1).permute(0, 3, 1, 2), xr.grad, dtype)
    check(name + ' dW vs torch', gw_g, wr.grad, 'f32' if dtype == 'f32' else 'bf16', scale=float(wr.grad.abs().max()))


SX_CASES = [c for c in CONV_CASES if ' tile' in c[0] or 'stem ragged' in c[0] or c[0] == '2x2 head co1']


@pytest.mark.parametrize('case', SX_CASES, ids=[c[0] for c in SX_CASES])
def test_strided_wgrad_takes_the_tile_kernel(case):
    """The strided cases above are served by conv_wgrad_sx_kernel, not by the general gather kernel: only the tile kernels split
    the pixel range into partial slabs (segnb_conv_wgrad_slabs > 1; the general kernel accumulates into one)."""
    name, N, H, W, segs, Co, k, s, p, transposed = case
    Ci = sum(r for r, _ in segs)
    rt = Runtime('cuda', 'bf16')
    w = torch.zeros((Ci, Co, k, k) if transposed else (Co, Ci, k, k), device='cuda')
    op = ConvOp(rt, w, None, segs, s, p, transposed, need_dgrad=True)
    assert len(SX_CASES) == 15
    assert min(op.plan(H, W)['nslab']) > 1, op.plan(H, W)['nslab']


DEEPK_CASES = [c for c in CONV_CASES if 'deep K' in c[0]]


@pytest.mark.parametrize('case', DEEPK_CASES, ids=[c[0] for c in DEEPK_CASES])
def test_conv_fprop_deepk(case):
    """conv_fprop_deepk_kernel (forward without statistics of few pixels x few channels x deep K: the FCDenseNet bottleneck
    layers, tiramisu.py:14) against the general kernel it replaces, the emulator and F.conv2d; reproducible run to run."""
    name, N, H, W, segs, Co, k, s, p, transposed = case
    Ci = sum(r for r, _ in segs)
    gen = torch.Generator().manual_seed(hash(name) % 1000)
    w = (torch.randn((Co, Ci, k, k), generator=gen) * (2.0 / (Ci * k * k)) ** 0.5).bfloat16().float()
    b = torch.randn(Co, generator=gen) * 0.1
    x = torch.randn(N, Ci, H, W, generator=gen).bfloat16().float()

    def run(device):
        rt = Runtime(device, 'bf16')
        op = ConvOp(rt, w.to(device), b.to(device), segs, s, p, transposed, need_dgrad=False)
        op.pack(H, W)
        Ho, Wo = op.out_hw(H, W)
        ld_in = op.Cip + 8
        xbuf = rt.zeros((N, H, W, ld_in))
        xv = View(xbuf, N, H, W, op.Cip, ld_in, 8)
        off, roff = 0, 0
        for real, padded in segs:
            xv.dense()[..., off:off + real] = x[:, roff:roff + real].permute(0, 2, 3, 1).to(device, rt.tdtype)
            off += padded
            roff += real
        ybuf = rt.zeros((N, Ho, Wo, op.Cop + 16))
        ybuf.fill_(7.0)
        yv = View(ybuf, N, Ho, Wo, op.Cop, op.Cop + 16, 8)       # a channel slice of a wider buffer
        op.fprop(xv, yv, None)
        if device != 'cpu':
            torch.cuda.synchronize()
        return yv.dense().float().cpu(), ybuf.float().cpu()

    outs = {}
    for knob in (1, 0, 1):
        nv.call('segnb_tune', b'fprop_deepk', knob)
        try:
            outs.setdefault(knob, []).append(run('cuda'))
        finally:
            nv.call('segnb_tune', b'fprop_deepk', 1)
    (y1, buf1), (y1b, _) = outs[1]
    y0, _ = outs[0][0]
    assert torch.equal(y1, y1b)                                      # fixed summation order
    assert float(buf1[..., :8].min()) == 7.0 and float(buf1[..., 8 + y1.shape[-1]:].min()) == 7.0     # nothing outside the slice
    with on_emulator():
        ye, _ = run('cpu')
    check(name + ' deepk vs general', y1, y0, 'bf16')
    check(name + ' deepk vs emulator', y1, ye, 'bf16')
    yr = F.conv2d(x, w, b, stride=s, padding=p)
    check(name + ' deepk vs torch', y1[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
    assert float(y1[..., Co:].abs().max()) == 0.0 if y1.shape[-1] > Co else True


# ------------------------------------------------------------------------------------------------------
# BN + activation + dropout + pool + upsample, forward and backward
# ------------------------------------------------------------------------------------------------------
def _run_bn(device, dtype, N, H, W, C, act, use_pool, use_up, use_drop, tensors):
    rt = Runtime(device, dtype)
    Cp = cp.pad8(C)
    y, gamma, beta, gd, gp, gu, drop = tensors
    dev = rt.device
    yv = View.alloc(rt, N, H, W, Cp)
    yv.dense()[..., :C] = y.to(dev, rt.tdtype)
    stats = torch.zeros(16, 2, Cp, dtype=torch.float64, device=dev)
    yy = yv.dense().double()
    stats[3, 0] = yy.sum((0, 1, 2))        # any replica: the finalize sums them
    stats[11, 1] = (yy * yy).sum((0, 1, 2))
    coef = rt.zeros((4, Cp), torch.float32)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    g_, b_ = gamma.to(dev), beta.to(dev)
    nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * H * W), nv.ptr(g_), nv.ptr(b_), 1e-5, 0.1,
            nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), 1, nv.ptr(coef), rt.stream)
    dm = None
    if use_drop:
        dm = torch.ones(N, Cp, device=dev)
        dm[:, :C] = drop.to(dev)
    out = View.alloc(rt, N, H, W, Cp)
    pool = View.alloc(rt, N, H // 2, W // 2, Cp) if use_pool else None
    up = View.alloc(rt, N, 2 * H, 2 * W, Cp) if use_up else None
    nv.call('segnb_bn_act_fwd', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm), out.ptr,
            out.ld, None if pool is None else pool.ptr, 0 if pool is None else pool.ld,
            None if up is None else up.ptr, 0 if up is None else up.ld, None, 0, rt.stream)
    gdv = View.alloc(rt, N, H, W, Cp)
    gdv.dense()[..., :C] = gd.to(dev, rt.tdtype)
    gpv = gupv = None
    if use_pool:
        gpv = View.alloc(rt, N, H // 2, W // 2, Cp)
        gpv.dense()[..., :C] = gp.to(dev, rt.tdtype)
    if use_up:
        gupv = View.alloc(rt, N, 2 * H, 2 * W, Cp)
        gupv.dense()[..., :C] = gu.to(dev, rt.tdtype)
    dz = View.alloc(rt, N, H, W, Cp)
    sums = rt.zeros((16, 2, Cp), torch.float64)
    nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm),
            gdv.ptr, gdv.ld, None if gpv is None else gpv.ptr, 0 if gpv is None else gpv.ld,
            None if gupv is None else gupv.ptr, 0 if gupv is None else gupv.ld, dz.ptr, dz.ld, nv.ptr(sums),
            None, 0, rt.stream)
    sums_copy = sums.sum(0)
    bcoef = rt.zeros((3, Cp), torch.float32)
    dgam, dbet, dbias = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, float(N * H * W), nv.ptr(g_), nv.ptr(coef), nv.ptr(bcoef),
            nv.ptr(dgam), nv.ptr(dbet), 0, rt.stream)
    dyv = View.alloc(rt, N, H, W, Cp)
    nv.call('segnb_bn_bwd_apply', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), dz.ptr, dz.ld,
            dyv.ptr, dyv.ld, nv.ptr(dbias), C, rt.stream)
    if device != 'cpu':
        torch.cuda.synchronize()
    res = dict(coef=coef, rm=rm, rv=rv, nbt=nbt.float(), out=out.dense(), dz=dz.dense(), sums=sums_copy,
               bcoef=bcoef, dgam=dgam, dbet=dbet, dy=dyv.dense(), dbias=dbias, stats_after=stats, sums_after=sums)
    if pool is not None:
        res['pool'] = pool.dense()
    if up is not None:
        res['up'] = up.dense()
    return {k: v.detach().double().cpu() for k, v in res.items()}


BN_CASES = [
    # N, H,  W,  C,   act,          pool,  up,    drop
    (2, 8,  8,  32,  nv.ACT_RELU,  True,  False, True),
    (3, 7,  9,  20,  nv.ACT_RELU,  True,  False, False),     # odd sizes, padded channels (20 -> 24)
    (2, 6,  10, 96,  nv.ACT_RELU,  False, True,  True),      # 12 chunks: not a power of two
    (2, 12, 12, 264, nv.ACT_LEAKY, False, False, False),     # > 32 chunks: two chunk tiles
    (1, 16, 16, 8,   nv.ACT_NONE,  True,  True,  False),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', BN_CASES, ids=[str(c[:4]) for c in BN_CASES])
def test_bn_act_pool_upsample_fwd_bwd(case, dtype):
    N, H, W, C, act, use_pool, use_up, use_drop = case
    gen = torch.Generator().manual_seed(C + H)
    y = torch.randn(N, H, W, C, generator=gen) * 2 + 0.5
    tensors = (y, 1 + 0.3 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen),
               torch.randn(N, H, W, C, generator=gen), torch.randn(N, H // 2, W // 2, C, generator=gen),
               torch.randn(N, 2 * H, 2 * W, C, generator=gen),
               (torch.rand(N, C, generator=gen) > 0.3).float() / 0.7)
    g = _run_bn('cuda', dtype, N, H, W, C, act, use_pool, use_up, use_drop, tensors)
    with on_emulator():
        e = _run_bn('cpu', dtype, N, H, W, C, act, use_pool, use_up, use_drop, tensors)
    for k in ('coef', 'rm', 'rv', 'nbt', 'bcoef'):
        check(k, g[k], e[k], 'f32')
    assert float(g['stats_after'].abs().max()) == 0.0 and float(g['sums_after'].abs().max()) == 0.0
    for k in ('out', 'pool', 'up', 'dz', 'dy'):
        if k in g:
            check(k, g[k], e[k], dtype)
    for k in ('sums', 'dgam', 'dbet'):
        check(k, g[k], e[k], 'f32' if dtype == 'f32' else 'bf16')
    check('dbias', g['dbias'], e['dbias'], dtype, scale=float(e['dz'].abs().sum((0, 1, 2)).max()))
    if dtype == 'f32' and act == nv.ACT_RELU and not use_up:
        # independent reference: torch autograd through BatchNorm2d(train) -> ReLU -> Dropout2d table -> pool
        yt = y.permute(0, 3, 1, 2).clone().requires_grad_(True)
        gam, bet = tensors[1].clone().requires_grad_(True), tensors[2].clone().requires_grad_(True)
        a = torch.relu(F.batch_norm(yt, None, None, gam, bet, True, 0.1, 1e-5))
        if use_drop:
            a = a * tensors[6][:, :, None, None]
        loss = (a * tensors[3].permute(0, 3, 1, 2)).sum()
        if use_pool:
            loss = loss + (F.max_pool2d(a, 2) * tensors[4].permute(0, 3, 1, 2)).sum()
        loss.backward()
        check('out vs torch', g['out'][..., :C].permute(0, 3, 1, 2), a, 'f32')
        check('dy vs torch', g['dy'][..., :C].permute(0, 3, 1, 2), yt.grad, 'f32')
        check('dgamma vs torch', g['dgam'], gam.grad, 'f32')
        check('dbeta vs torch', g['dbet'], bet.grad, 'f32')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', BN_CASES[:4], ids=[str(c[:4]) for c in BN_CASES[:4]])
def test_bn_fused_finalize_equals_separate_launches(case, dtype):
    """segnb_bn_fwd_fused / segnb_bn_bwd_apply_fused == finalize + pass, bit for bit; and the accumulator hand-over:
    the forward clears the backward sums, the backward clears the forward statistics."""
    N, H, W, C, act, use_pool, use_up, use_drop = case
    rt = Runtime('cuda', dtype)
    Cp = cp.pad8(C)
    gen = torch.Generator().manual_seed(C + H + 7)
    yv = _view_from(rt, torch.randn(N, H, W, C, generator=gen) * 2 + 0.5, Cp)
    gamma = (1 + 0.3 * torch.randn(C, generator=gen)).cuda()
    beta = (0.2 * torch.randn(C, generator=gen)).cuda()
    dm = None
    if use_drop:
        dm = torch.ones(N, Cp, device='cuda')
        dm[:, :C] = ((torch.rand(N, C, generator=gen) > 0.3).float() / 0.7).cuda()
    gd = _view_from(rt, torch.randn(N, H, W, C, generator=gen), Cp)
    res = {}
    for fused in (False, True):
        stats = rt.zeros((16, 2, Cp), torch.float64)
        nv.call('segnb_bn_stats', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(stats), rt.stream)
        stats0 = stats.clone()
        sums = rt.zeros((16, 2, Cp), torch.float64)
        sums.fill_(123.0 if fused else 0.0)         # the fused forward must clear them
        coef, bcoef = rt.zeros((4, Cp), torch.float32), rt.zeros((3, Cp), torch.float32)
        rm, rvv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        nbt = torch.zeros((), dtype=torch.int64, device='cuda')
        out = View.alloc(rt, N, H, W, Cp)
        pool = View.alloc(rt, N, H // 2, W // 2, Cp) if use_pool else None
        up = View.alloc(rt, N, 2 * H, 2 * W, Cp) if use_up else None
        tail = (act, 0.01, nv.ptr(dm), out.ptr, out.ld, None if pool is None else pool.ptr,
                0 if pool is None else pool.ld, None if up is None else up.ptr, 0 if up is None else up.ld, None, 0,
                rt.stream)
        if fused:
            nv.call('segnb_bn_fwd_fused', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(stats), nv.ptr(gamma),
                    nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm), nv.ptr(rvv), nv.ptr(nbt), nv.ptr(coef), nv.ptr(sums), *tail)
            assert torch.equal(stats, stats0) and float(sums.abs().max()) == 0.0
        else:
            nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * H * W), nv.ptr(gamma), nv.ptr(beta), 1e-5, 0.1,
                    nv.ptr(rm), nv.ptr(rvv), nv.ptr(nbt), 1, nv.ptr(coef), rt.stream)
            nv.call('segnb_bn_act_fwd', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), *tail)
        dz = View.alloc(rt, N, H, W, Cp)
        nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm),
                gd.ptr, gd.ld, None, 0, None, 0, dz.ptr, dz.ld, nv.ptr(sums), None, 0, rt.stream)
        dgam, dbet = torch.ones(C, device='cuda'), torch.ones(C, device='cuda')     # accumulate on top of 1
        dy = View.alloc(rt, N, H, W, Cp)
        if fused:
            nv.call('segnb_bn_bwd_apply_fused', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums),
                    nv.ptr(gamma), nv.ptr(bcoef), nv.ptr(dgam), nv.ptr(dbet), 1, nv.ptr(stats), dz.ptr, dz.ld, dy.ptr,
                    dy.ld, rt.stream)
            assert float(stats.abs().max()) == 0.0
            # the accumulating form (gradient of a multi-consumer input): old + result == segnb_add of the two
            prior = _view_from(rt, torch.randn(N, H, W, C, generator=gen), Cp)
            want = View.alloc(rt, N, H, W, Cp)
            nv.call('segnb_add', rt.code, prior.ptr, prior.ld, dy.ptr, dy.ld, want.ptr, want.ld, N, H, W, Cp, rt.stream)
            dg2, db2, bc2 = torch.ones(C, device='cuda'), torch.ones(C, device='cuda'), rt.zeros((3, Cp), torch.float32)
            nv.call('segnb_bn_bwd_apply_fused_acc', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums),
                    nv.ptr(gamma), nv.ptr(bc2), nv.ptr(dg2), nv.ptr(db2), 1, None, dz.ptr, dz.ld, prior.ptr, prior.ld,
                    rt.stream)
            if not use_drop:
                # ... and with dz recomputed from the incoming gradient (no dz tensor): the same sums, the same result
                prior2 = _view_from(rt, prior.dense()[..., :C].float().cpu() * 0 + 1.0, Cp)
                want2 = View.alloc(rt, N, H, W, Cp)
                nv.call('segnb_add', rt.code, prior2.ptr, prior2.ld, dy.ptr, dy.ld, want2.ptr, want2.ld, N, H, W, Cp, rt.stream)
                dg3, db3, bc3 = torch.ones(C, device='cuda'), torch.ones(C, device='cuda'), rt.zeros((3, Cp), torch.float32)
                nv.call('segnb_bn_bwd_apply_fused_direct_acc', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef),
                        nv.ptr(sums), nv.ptr(gamma), nv.ptr(bc3), nv.ptr(dg3), nv.ptr(db3), 1, None, act, 0.01, gd.ptr, gd.ld,
                        prior2.ptr, prior2.ld, rt.stream)
                torch.cuda.synchronize()
                err2 = float((prior2.dense().float() - want2.dense().float()).abs().max())
                assert err2 == 0.0 if dtype == 'bf16' else err2 <= 2.5e-7 * float(want2.dense().abs().max()), err2
                assert torch.equal(dg3, dgam) and torch.equal(bc3, bcoef)
            torch.cuda.synchronize()
            assert torch.equal(dg2, dgam) and torch.equal(bc2, bcoef)
            # (bf16: bit for bit; fp32: the two instantiations may contract the BatchNorm expression differently -- one ulp)
            err = float((prior.dense().float() - want.dense().float()).abs().max())
            assert err == 0.0 if dtype == 'bf16' else err <= 2.5e-7 * float(want.dense().abs().max()), err
        else:
            nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, float(N * H * W), nv.ptr(gamma), nv.ptr(coef),
                    nv.ptr(bcoef), nv.ptr(dgam), nv.ptr(dbet), 1, rt.stream)
            nv.call('segnb_bn_bwd_apply', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), dz.ptr,
                    dz.ld, dy.ptr, dy.ld, None, C, rt.stream)
        torch.cuda.synchronize()
        res[fused] = dict(coef=coef, bcoef=bcoef, rm=rm, rv=rvv, nbt=nbt, out=out.dense(), dy=dy.dense(), dgam=dgam,
                          dbet=dbet, pool=None if pool is None else pool.dense(), up=None if up is None else up.dense())
    for k, v in res[False].items():
        if v is not None:
            assert torch.equal(v, res[True][k]), k


# ------------------------------------------------------------------------------------------------------
# head, losses, SGD, input packing
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 16, 16, 32, 1), (1, 9, 13, 20, 3), (2, 8, 8, 48, 2),
                                   (2, 12, 20, 256, 1), (1, 7, 9, 272, 5), (2, 6, 10, 72, 2)])     # > 64 channels: the wide forward kernel
def test_head_fwd_bwd(shape, dtype):
    N, H, W, C, K = shape
    Cp = cp.pad8(C)
    gen = torch.Generator().manual_seed(C)
    a = torch.randn(N, H, W, C, generator=gen)
    w = torch.randn(K, C, 1, 1, generator=gen) * 0.2
    b = torch.randn(K, generator=gen)
    dl = torch.randn(N, K, H, W, generator=gen)
    if dtype == 'bf16':
        a = a.bfloat16().float()

    def run(device):
        rt = Runtime(device, dtype)
        av = View.alloc(rt, N, H, W, Cp)
        av.dense()[..., :C] = a.to(rt.device, rt.tdtype)
        wd, bd, dld = w.to(rt.device), b.to(rt.device), dl.to(rt.device)
        logits = torch.zeros(N, K, H, W, device=rt.device)
        nv.call('segnb_head_fwd', rt.code, av.ptr, av.ld, N, H, W, C, nv.ptr(wd), nv.ptr(bd), K, nv.ptr(logits),
                rt.stream)
        da = View.alloc(rt, N, H, W, Cp)
        dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
        nv.call('segnb_head_bwd', rt.code, av.ptr, av.ld, N, H, W, C, Cp, nv.ptr(wd), K, nv.ptr(dld), da.ptr, da.ld,
                nv.ptr(dw), nv.ptr(db), rt.stream)
        if device != 'cpu':
            torch.cuda.synchronize()
        return logits.cpu(), da.dense().float().cpu(), dw.cpu(), db.cpu()

    lg, dag, dwg, dbg = run('cuda')
    with on_emulator():
        le, dae, dwe, dbe = run('cpu')
    check('logits', lg, le, 'f32')
    check('da', dag, dae, dtype)
    check('dw', dwg, dwe, 'f32')
    check('db', dbg, dbe, 'f32')
    ar = a.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    out = F.conv2d(ar, wr, b)
    out.backward(dl)
    check('logits vs torch', lg, out, 'f32')
    check('da vs torch', dag[..., :C].permute(0, 3, 1, 2), ar.grad, dtype)
    check('dw vs torch', dwg, wr.grad, 'f32')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('act', [-1, nv.ACT_RELU, nv.ACT_LEAKY])
@pytest.mark.parametrize('shape', [(2, 21, 19, 32, 1, 2, 2, 1),      # linknet.py:62 finalconv3: Conv2d(32, 1, 2, padding=1)
                                   (2, 16, 24, 32, 1, 1, 1, 0),      # unet16.py:111 final: 1 x 1
                                   (1, 9, 13, 20, 2, 2, 2, 1), (1, 11, 10, 64, 8, 1, 1, 0), (2, 8, 9, 16, 2, 1, 3, 1),
                                   (1, 12, 12, 24, 1, 2, 2, 0)])
def test_head_conv_fwd_bwd(shape, act, dtype):
    """The classifier-as-a-small-convolution kernels against torch (forward, da, dw, db) and against the emulator; with act the
    backward also applies the producing layer's activation mask and sums dz (that layer's bias gradient)."""
    N, H, W, C, K, kh, kw, pad = shape
    Cp = cp.pad8(C)
    assert nv.query('segnb_head_conv_ok', C, K, kh, kw)
    gen = torch.Generator().manual_seed(C + kh)
    a = torch.randn(N, H, W, C, generator=gen)
    w = torch.randn(K, C, kh, kw, generator=gen) * 0.2
    b = torch.randn(K, generator=gen)
    Ho, Wo = H + 2 * pad - kh + 1, W + 2 * pad - kw + 1
    dl = torch.randn(N, K, Ho, Wo, generator=gen)
    if dtype == 'bf16':
        a = a.bfloat16().float()
    slope = 0.01

    def run(device):
        rt = Runtime(device, dtype)
        av = View.alloc(rt, N, H, W, Cp)
        av.dense()[..., :C] = a.to(rt.device, rt.tdtype)
        wd, bd, dld = w.to(rt.device), b.to(rt.device), dl.to(rt.device)
        logits = torch.zeros(N, K, Ho, Wo, device=rt.device)
        nv.call('segnb_head_conv_fwd', rt.code, av.ptr, av.ld, N, H, W, C, nv.ptr(wd), kh, kw, pad, nv.ptr(bd), K, nv.ptr(logits),
                rt.stream)
        da = View.alloc(rt, N, H, W, Cp)
        da.dense().fill_(7.0)
        dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
        sums = torch.zeros(16, 2, Cp, dtype=torch.float64, device=rt.device)
        nv.call('segnb_head_conv_bwd', rt.code, av.ptr, av.ld, N, H, W, C, Cp, nv.ptr(wd), kh, kw, pad, K, nv.ptr(dld), act, slope,
                da.ptr, da.ld, nv.ptr(dw), nv.ptr(db), nv.ptr(sums) if act >= 0 else None, rt.stream)
        if device != 'cpu':
            torch.cuda.synchronize()
        return logits.cpu(), da.dense().float().cpu(), dw.cpu(), db.cpu(), sums.sum(0).cpu()

    lg, dag, dwg, dbg, sg = run('cuda')
    with on_emulator():
        le, dae, dwe, dbe, se = run('cpu')
    check('logits', lg, le, 'f32')
    check('da', dag, dae, dtype)
    check('dw', dwg, dwe, 'f32')
    check('db', dbg, dbe, 'f32')
    ar = a.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    out = F.conv2d(ar, wr, br, padding=pad)
    out.backward(dl)
    check('logits vs torch', lg, out, 'f32')
    g = ar.grad.permute(0, 2, 3, 1)
    if act >= 0:
        g = torch.where(a > 0, g, g * (0.0 if act == nv.ACT_RELU else slope))
    check('da vs torch', dag[..., :C], g, dtype)
    assert float(dag[..., C:].abs().max()) == 0.0 if Cp > C else True
    check('dw vs torch', dwg, wr.grad, 'f32')
    check('db vs torch', dbg, br.grad, 'f32')
    if act >= 0:
        # the sums are those of the STORED dz values
        check('sums', sg[0].float(), dag.double().reshape(-1, Cp).sum(0).float(), 'f32', scale=float(dag.abs().sum(dim=(0, 1, 2)).max()))
        assert float(sg[1].abs().max()) == 0.0


LOSS_NAMES = ['bce', 'jaccard', 'smooth_jaccard', 'dice', 'bce_jaccard', 'bce_dice', 'focal']


@pytest.mark.parametrize('name', LOSS_NAMES)
def test_losses_vs_reference_golden(golden_dir, name):
    """HIP loss kernels vs values + gradients produced by the REFERENCE's lib/losses.py (fixture)."""
    import os
    from lib import losses as L
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    t = torch.from_numpy(g['t']).cuda()
    crit = {'bce': L.BCEWithSigmoidLoss, 'jaccard': L.JaccardLoss, 'smooth_jaccard': L.SmoothJaccardLoss,
            'dice': L.DiceLoss, 'bce_jaccard': L.BCEWithLogitsLossAndSmoothJaccard, 'bce_dice': L.BCEAndDiceLoss,
            'focal': lambda: L.FocalLossBinary(size_average=False)}[name]()
    l = crit(x, t)
    (x.shape[0] * l).backward()
    assert abs(l.item() - float(g['loss_' + name])) <= 1e-5 * max(1.0, abs(float(g['loss_' + name])))
    ref = g['dx_' + name]
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-4, atol=2e-6 * np.abs(ref).max())


def test_metrics_vs_reference_golden(golden_dir):
    import os
    from lib.metrics import JaccardScore, PixelAccuracy
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    x, t = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['t']).cuda()
    assert abs(JaccardScore()(x, t).item() - float(g['iou'])) < 1e-6
    assert abs(PixelAccuracy()(x, t).item() - float(g['acc'])) < 1e-7
    xe = torch.full((1, 1, 2, 2), 3.0).cuda()
    assert PixelAccuracy()(xe, torch.zeros(1, 1, 2, 2).long().cuda()).item() == 0.0


@pytest.mark.parametrize('tag', __import__('model_checks').EXTRA_LOSS_CASES)
def test_loss_constructor_surface_vs_reference_golden(golden_dir, tag):
    """BCEWithSigmoidLoss(size_average=False | reduce=False), FocalLossBinary(gamma != 2) on the HIP kernels vs values
    and gradients produced by the REFERENCE's lib/losses.py:47-53,84-101."""
    import os
    import model_checks as mc
    mc.check_extra_loss_case(np.load(os.path.join(golden_dir, 'losses.npz')), tag, 'cuda')


def test_metrics_come_from_the_loss_sums(golden_dir):
    """JaccardScore / PixelAccuracy right after the loss on the same tensors (torch_train.py:185,209-210): answered
    from the loss launch's sums -- same values as the stand-alone pass and as the reference."""
    import os
    from lib.losses import BCEWithLogitsLossAndSmoothJaccard
    from lib.metrics import JaccardScore, PixelAccuracy
    from segnb import seglosses
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    x, t = torch.from_numpy(g['x']).cuda().requires_grad_(True), torch.from_numpy(g['t']).cuda()
    BCEWithLogitsLossAndSmoothJaccard()(x, t)
    assert seglosses._recall(x, t) is not None
    iou, acc = JaccardScore()(x, t).item(), PixelAccuracy()(x, t).item()
    assert abs(iou - float(g['iou'])) < 1e-6 and abs(acc - float(g['acc'])) < 1e-7
    x2 = x.detach().clone()
    assert seglosses._recall(x2, t) is None
    assert JaccardScore()(x2, t).item() == iou


def test_grad_absmax_kernel():
    gen = torch.Generator().manual_seed(5)
    for n in (4, 1003, 31454724):
        v = torch.randn(n, generator=gen)
        v[n // 3] = -7.25 if n > 4 else 0.5
        d = v.cuda()
        out = torch.zeros((), device='cuda')
        nv.call('segnb_absmax_f32', nv.ptr(d), n, nv.ptr(out), torch.cuda.current_stream().cuda_stream)
        assert out.item() == v.abs().max().item()
    z = torch.zeros(64, device='cuda')
    out = torch.ones((), device='cuda')
    nv.call('segnb_absmax_f32', nv.ptr(z), 64, nv.ptr(out), torch.cuda.current_stream().cuda_stream)
    assert out.item() == 0.0


def test_pr_curve_histogram_kernel_vs_threshold_loop():
    """PRCurveMeter.update on the GPU (segnb_pr_histogram) vs the reference's 127-threshold loop
    (lib/train_utils.py:109-125) restated in numpy.  A pixel whose probability sits within 2e-7 of a threshold may
    fall on either side (sigmoid implementations differ by an ulp): counts must agree up to those pixels."""
    from lib.train_utils import PRCurveMeter
    gen = torch.Generator().manual_seed(0)
    logits = 3 * torch.randn(4, 1, 224, 224, generator=gen)
    y = (torch.rand(4, 1, 224, 224, generator=gen) > 0.6).long()
    m = PRCurveMeter()
    m.update(logits.cuda(), y.cuda())
    p = torch.sigmoid(logits.double()).numpy().reshape(-1)
    t = y.numpy().reshape(-1).astype(np.int64)
    for i, v in enumerate(np.arange(0., 1., 1. / 127, dtype=np.float32)):
        pred = (p > float(v)).astype(np.int64)
        conf = np.bincount(pred + 2 * t, minlength=4).reshape(2, 2)
        slack = int((np.abs(p - float(v)) < 2e-7).sum())
        for got, ref in ((m.tp[i], conf[1, 1]), (m.tn[i], conf[0, 0]), (m.fp[i], conf[0, 1]), (m.fn[i], conf[1, 0])):
            assert abs(int(got) - int(ref)) <= slack, (i, got, ref, slack)
    assert int(m.tp[0] + m.fn[0]) == int(t.sum()) and int(m.tp[5] + m.tn[5] + m.fp[5] + m.fn[5]) == t.size


def test_snapshot_round_trip_gpu(tmp_path):
    """save_snapshot / restore_snapshot (torch_train.py:308-330) with the bf16 HIP path and the one-launch Adam: model
    and optimizer state come back bit for bit; the resumed run then tracks the uninterrupted one (not bitwise: the head's
    weight gradient is accumulated with fp32 atomics, and Adam turns an ulp of a near-zero gradient into lr-sized steps)."""
    import torch_train as TT
    from lib.models.zf_unet import ZF_UNET
    torch.manual_seed(0)
    model = ZF_UNET(filters=8, dropout_val=0.0).cuda()
    opt = TT.get_optimizer('adam', model.parameters(), 1e-3)
    gen = torch.Generator().manual_seed(1)
    data = [(torch.randn(4, 3, 64, 64, generator=gen), (torch.rand(4, 1, 64, 64, generator=gen) > 0.7).long())
            for _ in range(2)]
    seen = []
    TT.train(model, TT.get_loss('bce_jaccard'), opt, data, metrics=TT.default_metrics(),
             grad_monitor=lambda step, v: seen.append(v))
    assert seen[-1] == max(p.grad.abs().max().item() for p in model.parameters())
    f = str(tmp_path / 'snap.pth')
    TT.save_snapshot(model, opt, 0.25, 0, {'epoch': {0: 0}}, f)
    m2 = ZF_UNET(filters=8, dropout_val=0.0).cuda()
    o2 = TT.get_optimizer('adam', m2.parameters(), 1e-3)
    start, _, best = TT.restore_snapshot(m2, o2, f)
    assert start == 1 and best == 0.25
    for (k, a), b in zip(model.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    sa, sb = opt.state_dict()['state'], o2.state_dict()['state']
    assert len(sa) == len(sb) == len(list(model.parameters()))
    for i in sa:
        assert float(sa[i]['step']) == float(sb[i]['step']) == 2.0
        assert torch.equal(sa[i]['exp_avg'], sb[i]['exp_avg']) and torch.equal(sa[i]['exp_avg_sq'], sb[i]['exp_avg_sq'])
    la, _ = TT.train(model, TT.get_loss('bce_jaccard'), opt, data)
    lb, _ = TT.train(m2, TT.get_loss('bce_jaccard'), o2, data)
    assert abs(la.avg - lb.avg) < 1e-3
    for (k, a), b in zip(model.named_parameters(), m2.parameters()):
        assert float((a - b).abs().max()) <= 5e-3, k            # two Adam steps of lr 1e-3 each
    assert float(o2.state_dict()['state'][0]['step']) == 4.0


def test_sgd_and_input_pack():
    gen = torch.Generator().manual_seed(0)
    p = torch.randn(1000003 // 4 * 4 + 3, generator=gen)
    g = torch.randn(p.numel(), generator=gen)
    pd, gd = p.cuda(), g.cuda()
    nv.call('segnb_sgd_step', nv.ptr(pd), nv.ptr(gd), p.numel(), 0.125, torch.cuda.current_stream().cuda_stream)
    torch.testing.assert_close(pd.cpu(), p - 0.125 * g, rtol=1e-6, atol=1e-6)
    x = torch.randn(3, 3, 10, 14, generator=gen)
    for dtype, tdt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
        rt = Runtime('cuda', dtype)
        out = torch.full((3, 10, 14, 16), 7.0, dtype=tdt, device='cuda')
        nv.call('segnb_pack_input_nchw', nv.ptr(x.cuda()), 3, 3, 10, 14, nv.ptr(out), rt.code, 8, 16, rt.stream)
        torch.cuda.synchronize()
        ref = x.permute(0, 2, 3, 1).to(tdt).float()
        assert torch.equal(out[..., :3].float().cpu(), ref)
        assert float(out[..., 3:8].abs().max()) == 0.0 and float((out[..., 8:] - 7).abs().max()) == 0.0


def test_both_matrices_of_a_convolution_from_one_read():
    """segnb_pack_weight_pair_multi: the forward matrix [Cop][9][Cip] and the data-gradient matrix [Cip][9][Cop] of a plain
    nn.Conv2d(ci, co, 3) weight (lib/models/zf_unet.py:5-32) from ONE read of the parameter == the two single-form packs, bit
    for bit, incl. ragged tiles (co not a multiple of 32, ci not of 64), padded channels (written as zeros) and the widest
    layers of the timed configuration, rows that are not 16-byte aligned (3 and 5 input channels) and a layer without a data
    gradient; remapped channels (a padded concat) stay with the tiled pack."""
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(3)
    specs = [([(32, 32)], 32, True), ([(4, 8)], 8, True), ([(100, 104)], 70, True), ([(64, 64), (32, 32)], 40, True),
             ([(1024, 1024)], 512, True), ([(192, 192)], 200, True), ([(3, 8)], 32, True), ([(12, 16), (6, 8)], 16, False),
             ([(30, 32)], 24, True), ([(5, 8)], 16, True)]
    ops, pj = [], []
    for k, (segs, co, _) in enumerate(specs):
        ci = sum(r for r, _ in segs)
        w = torch.randn(co, ci, 3, 3, generator=gen).cuda()
        op = ConvOp(rt, w, None, segs, 1, 1, False, k != len(specs) - 1)      # (the last one: no data gradient, forward matrix only)
        op.plan(16, 16)
        ops.append(op)
        pj += op.pack_jobs(16, 16)
    mats = lambda op: op.plan(16, 16)['wp_fwd'] + op.plan(16, 16).get('wp_dg', [])
    keep = PackTable.pair_pack
    try:
        PackTable.pair_pack = False
        t0 = PackTable(rt, pj, 'segnb_pack_weight_multi', 'segnb_pack_weight')
        assert t0.pn == 0
        t0.run()
        torch.cuda.synchronize()
        ref = [[t.clone() for t in mats(op)] for op in ops]
        for op in ops:
            for t in mats(op):
                t.fill_(7.0)
        PackTable.pair_pack = True
        t1 = PackTable(rt, pj, 'segnb_pack_weight_multi', 'segnb_pack_weight')
        assert t1.pn == sum(1 for s in specs if s[2]), t1.pn
        t1.run()
        torch.cuda.synchronize()
    finally:
        PackTable.pair_pack = keep
    for op, r, spec in zip(ops, ref, specs):
        for t, rr in zip(mats(op), r):
            assert torch.equal(t, rr), spec


@pytest.mark.parametrize('dtype', DTYPES)
def test_batched_pack_unpack_equals_single_job_calls(dtype):
    """segnb_pack_weight_multi / segnb_unpack_wgrad_multi (LDS-tiled, one launch for many matrices) vs the
    single-job entry points, bit for bit, incl. padded concat channel maps and a 4x4 kernel (falls back)."""
    from segnb.engine import PackTable
    rt = Runtime('cuda', dtype)
    gen = torch.Generator().manual_seed(1)
    specs = [([(3, 8)], 32, 3, False), ([(12, 16), (6, 8)], 6, 3, False), ([(300, 304)], 70, 3, False),
             ([(64, 64), (32, 32)], 40, 3, False), ([(32, 32)], 32, 4, True)]
    ops, pj, uj, grads = [], [], [], []
    for segs, co, k, tr in specs:
        ci = sum(r for r, _ in segs)
        w = torch.randn((ci, co, k, k) if tr else (co, ci, k, k), generator=gen).cuda()
        op = ConvOp(rt, w, None, segs, 2 if tr else 1, 1, tr, True)
        op.direct_dw = False               # (this test is about the workspace -> gradient unpack kernels themselves)
        op.plan(16, 16)
        g = torch.randn(w.shape, generator=gen).cuda()
        ops.append(op)
        grads.append(g)
        pj += op.pack_jobs(16, 16)
        uj += op.unpack_jobs(16, 16, g)
    # pack: single-job reference
    for op in ops:
        op.pack(16, 16)
    ref = [[t.clone() for t in op.plan(16, 16)['wp_fwd'] + op.plan(16, 16).get('wp_dg', [])] for op in ops]
    for op in ops:
        for t in op.plan(16, 16)['wp_fwd'] + op.plan(16, 16).get('wp_dg', []):
            t.fill_(7.0)
    PackTable(rt, pj, 'segnb_pack_weight_multi', 'segnb_pack_weight').run()
    torch.cuda.synchronize()
    for op, r in zip(ops, ref):
        for t, rr in zip(op.plan(16, 16)['wp_fwd'] + op.plan(16, 16).get('wp_dg', []), r):
            assert torch.equal(t, rr)
    # unpack: workspace (slab 0) -> gradient (+=), workspace re-zeroed
    expect = []
    for op, g in zip(ops, grads):
        p = op.plan(16, 16)
        for d in p['dwp']:
            d.copy_(torch.randn(d.shape, generator=gen))
        saved = [d.clone() for d in p['dwp']]
        g0 = g.clone()
        for job in op.unpack_jobs(16, 16, g0):
            nv.call('segnb_unpack_wgrad', nv.ptr(job['packed']), nv.ptr(job['w']), job['Mp'], job['Cp'], job['ntaps'],
                    job['s_m'], job['s_c'], nv.int_array(job['tap_off']), nv.ptr(job['mmap']), nv.ptr(job['cmap']), 1,
                    rt.stream)
        expect.append(g0)
        for d, sv in zip(p['dwp'], saved):
            d.copy_(sv)
    PackTable(rt, uj, 'segnb_unpack_wgrad_multi', 'segnb_unpack_wgrad').run()
    torch.cuda.synchronize()
    for op, g, e in zip(ops, grads, expect):
        torch.testing.assert_close(g, e, rtol=0, atol=0)
        assert all(float(d[0].abs().max()) == 0.0 for d in op.plan(16, 16)['dwp'])     # slab 0 consumed


# ------------------------------------------------------------------------------------------------------
# kernels added for LinkNet34 / FCDenseNet / UNet16: add, bn_stats, general max-pool, NHWC->NCHW, residual BN
# ------------------------------------------------------------------------------------------------------
def _view_from(rt, t_nhwc, Cp, slack=0):
    """NHWC host tensor -> device View with padded channels inside a wider (ld = Cp + slack) buffer."""
    N, H, W, C = t_nhwc.shape
    buf = rt.zeros((N, H, W, Cp + slack), rt.tdtype)
    v = View(buf, N, H, W, Cp, Cp + slack, 0)
    v.dense()[..., :C] = t_nhwc.to(rt.device, rt.tdtype)
    return v


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 9, 7, 20), (1, 32, 32, 64), (3, 5, 5, 264)])
def test_add_bn_stats_nhwc_to_nchw(shape, dtype):
    N, H, W, C = shape
    rt = Runtime('cuda', dtype)
    Cp = cp.pad8(C)
    gen = torch.Generator().manual_seed(C)
    a, b = torch.randn(N, H, W, C, generator=gen), torch.randn(N, H, W, C, generator=gen)
    av, bv = _view_from(rt, a, Cp, slack=16), _view_from(rt, b, Cp)
    ov = _view_from(rt, torch.zeros(N, H, W, C), Cp, slack=8)
    nv.call('segnb_add', rt.code, av.ptr, av.ld, bv.ptr, bv.ld, ov.ptr, ov.ld, N, H, W, Cp, rt.stream)
    ar, br = a.to(rt.tdtype).float(), b.to(rt.tdtype).float()
    ref = (ar + br).to(rt.tdtype).float()
    got = ov.dense().float().cpu()
    assert torch.equal(got[..., :C], ref), 'add must be bit-exact (one rounding of an fp32 sum)'
    assert float(got[..., C:].abs().max()) == 0.0 if Cp > C else True
    # bn_stats: fp64 sums of the stored values over 16 replicas
    stats = rt.zeros((16, 2, Cp), torch.float64)
    nv.call('segnb_bn_stats', rt.code, av.ptr, av.ld, N, H, W, Cp, nv.ptr(stats), rt.stream)
    s = stats.sum(0).cpu()
    ad = ar.double().reshape(-1, C)
    # per-thread fp32 partial sums, fp64 across threads/blocks: 1e-6 of the absolute sums
    assert float((s[0, :C] - ad.sum(0)).abs().max()) <= 2e-6 * float(ad.abs().sum(0).max())
    assert float((s[1, :C] - (ad * ad).sum(0)).abs().max()) <= 2e-6 * float((ad * ad).sum(0).max())
    # NHWC -> NCHW fp32 (real channels only)
    out = rt.zeros((N, C, H, W), torch.float32)
    nv.call('segnb_nhwc_to_nchw_f32', rt.code, av.ptr, av.ld, N, H, W, C, nv.ptr(out), rt.stream)
    assert torch.equal(out.cpu(), ar.permute(0, 3, 1, 2))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 3, 2, 1), (1, 15, 11, 24, 3, 2, 1), (2, 8, 10, 16, 2, 2, 0),
                                  (1, 9, 9, 8, 3, 1, 1)])
def test_maxpool_general_fwd_bwd(case, dtype):
    N, H, W, C, k, s, p = case
    rt = Runtime('cuda', dtype)
    gen = torch.Generator().manual_seed(H * W)
    x = torch.randn(N, H, W, C, generator=gen).to(rt.tdtype).float()
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    pr = F.max_pool2d(xr, k, s, p)
    Ho, Wo = pr.shape[2:]
    g = torch.randn(N, Ho, Wo, C, generator=gen).to(rt.tdtype).float()
    pr.backward(g.permute(0, 3, 1, 2))
    xv, gv = _view_from(rt, x, C, slack=8), _view_from(rt, g, C)
    ov, dxv, dxi = View.alloc(rt, N, Ho, Wo, C), View.alloc(rt, N, H, W, C), View.alloc(rt, N, H, W, C)
    idx = torch.zeros((N, Ho, Wo, C), dtype=torch.uint8, device='cuda')
    nv.call('segnb_maxpool_fwd', rt.code, xv.ptr, xv.ld, N, H, W, C, k, s, p, ov.ptr, ov.ld, nv.ptr(idx), rt.stream)
    nv.call('segnb_maxpool_bwd', rt.code, xv.ptr, xv.ld, gv.ptr, gv.ld, N, H, W, C, k, s, p, dxv.ptr, dxv.ld, None,
            rt.stream)
    nv.call('segnb_maxpool_bwd', rt.code, xv.ptr, xv.ld, gv.ptr, gv.ld, N, H, W, C, k, s, p, dxi.ptr, dxi.ld,
            nv.ptr(idx), rt.stream)
    assert torch.equal(ov.dense().float().cpu().permute(0, 3, 1, 2), pr.detach())
    # overlapping windows sum several gradients into one input: one rounding (bf16) / summation order (f32)
    check('maxpool dx', dxv.dense().permute(0, 3, 1, 2), xr.grad, dtype)
    # the backward from the argmax positions the forward recorded == the re-scanning one, bit for bit
    assert torch.equal(dxi.t, dxv.t)
    _, ir = F.max_pool2d(x.permute(0, 3, 1, 2), k, s, p, return_indices=True)
    a = ir // W - (torch.arange(Ho)[:, None] * s - p)
    b = ir % W - (torch.arange(Wo)[None, :] * s - p)
    assert torch.equal(idx.cpu().long(), (a * k + b).permute(0, 2, 3, 1))


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_act_with_residual_input(dtype):
    """BasicBlock tail: relu(bn(y) + res); dz is the gradient of both the BN output and the residual branch."""
    N, H, W, C = 2, 10, 6, 40
    rt = Runtime('cuda', dtype)
    gen = torch.Generator().manual_seed(5)
    q = lambda t: t.to(rt.tdtype).float()
    y, r, gd = (q(torch.randn(N, H, W, C, generator=gen)) for _ in range(3))
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    yv, rv, gv = _view_from(rt, y, C), _view_from(rt, r, C, slack=8), _view_from(rt, gd, C)
    stats = rt.zeros((16, 2, C), torch.float64)
    nv.call('segnb_bn_stats', rt.code, yv.ptr, yv.ld, N, H, W, C, nv.ptr(stats), rt.stream)
    coef = rt.zeros((4, C), torch.float32)
    rm, rvv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    nbt = torch.zeros((), dtype=torch.int64, device='cuda')
    g_, b_ = gamma.cuda(), beta.cuda()
    nv.call('segnb_bn_finalize', nv.ptr(stats), C, C, float(N * H * W), nv.ptr(g_), nv.ptr(b_), 1e-5, 0.1,
            nv.ptr(rm), nv.ptr(rvv), nv.ptr(nbt), 1, nv.ptr(coef), rt.stream)
    out, dz = View.alloc(rt, N, H, W, C), View.alloc(rt, N, H, W, C)
    nv.call('segnb_bn_act_fwd', rt.code, yv.ptr, yv.ld, N, H, W, C, nv.ptr(coef), nv.ACT_RELU, 0.0, None, out.ptr,
            out.ld, None, 0, None, 0, rv.ptr, rv.ld, rt.stream)
    sums = rt.zeros((16, 2, C), torch.float64)
    nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, N, H, W, C, nv.ptr(coef), nv.ACT_RELU, 0.0, None,
            gv.ptr, gv.ld, None, 0, None, 0, dz.ptr, dz.ld, nv.ptr(sums), rv.ptr, rv.ld, rt.stream)
    # torch reference
    yt = y.permute(0, 3, 1, 2).clone().requires_grad_(True)
    rt_ = r.permute(0, 3, 1, 2).clone().requires_grad_(True)
    a = torch.relu(F.batch_norm(yt, None, None, gamma, beta, True, 0.1, 1e-5) + rt_)
    a.backward(gd.permute(0, 3, 1, 2))
    check('res out', out.dense().permute(0, 3, 1, 2), a, dtype)
    # dz == d/d(res); elements within rounding of the ReLU kink may flip: compare where |pre-activation| is clear
    pre = (F.batch_norm(yt, None, None, gamma, beta, True, 0.1, 1e-5) + rt_).detach()
    clear = pre.abs() > (0.05 if dtype == 'bf16' else 1e-4)
    got = dz.dense().float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(got[clear], rt_.grad[clear])
    S = sums.sum(0).cpu()
    assert abs(float(S[0].sum()) - float(rt_.grad.double().sum())) <= (0.05 if dtype == 'bf16' else 1e-3) * \
        float(rt_.grad.abs().sum()) / 10


@pytest.mark.parametrize('name', ['rms', 'adam'])
def test_flat_rmsprop_adam_kernels_vs_torch(name):
    gen = torch.Generator().manual_seed(11)
    n = 100003
    p0 = torch.randn(n, generator=gen)
    ref_p = p0.clone().requires_grad_(True)
    ref = (torch.optim.RMSprop if name == 'rms' else torch.optim.Adam)([ref_p], lr=1e-2)
    p = p0.cuda()
    s1, s2 = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 4):
        g = torch.randn(n, generator=gen) * (10.0 ** torch.randint(-3, 2, (n,), generator=gen).float())
        ref_p.grad = g.clone()
        ref.step()
        gd = g.cuda()
        if name == 'rms':
            nv.call('segnb_rmsprop_step', nv.ptr(p), nv.ptr(gd), nv.ptr(s1), n, 1e-2, 0.99, 1e-8, 0)
        else:
            nv.call('segnb_adam_step', nv.ptr(p), nv.ptr(gd), nv.ptr(s1), nv.ptr(s2), n, 1e-2, 0.9, 0.999, 1e-8, step, 0)
        torch.cuda.synchronize()
        assert torch.allclose(p.cpu(), ref_p.detach(), rtol=2e-5, atol=1e-6), (name, step)


# ------------------------------------------------------------------------------------------------------
# direct-to-LDS 3x3 pipeline (fprop_dma.hip): every tile configuration, ragged tiles, sliced tensors
# ------------------------------------------------------------------------------------------------------
DMA_CASES = [
    # name,            N, H,  W,  segs,                     Co
    ('dma 64->64',     2, 21, 37, [(64, 64)],               64),      # ragged rows and columns of every tile shape (cfg -1 = automatic)
    ('dma 128->136',   1, 28, 28, [(128, 128)],             136),     # two channel tiles, the second mostly padding
    ('dma cat 192->64', 3, 14, 14, [(128, 128), (64, 64)],  64),      # three channel chunks (tile seams in the ring)
    ('dma 64->24',     5, 9,  50, [(64, 64)],               24),      # more tiles than one block round, thin output
]


@pytest.mark.parametrize('cfg', [-1, 0, 1])
@pytest.mark.parametrize('case', DMA_CASES, ids=[c[0] for c in DMA_CASES])
def test_conv_fprop_dma_configs(case, cfg):
    name, N, H, W, segs, Co = case
    full = (name, N, H, W, segs, Co, 3, 1, 1, False)
    Ci = sum(r for r, _ in segs)
    gen = torch.Generator().manual_seed(7 + cfg)
    w = (torch.randn((Co, Ci, 3, 3), generator=gen) * (2.0 / (Ci * 9)) ** 0.5).bfloat16().float()
    b = torch.randn(Co, generator=gen) * 0.1
    x = torch.randn(N, Ci, H, W, generator=gen).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=gen).bfloat16().float()
    nv.call('segnb_tune', b'fprop_dma', 1)
    nv.call('segnb_tune', b'fprop_dma_cfg', cfg)
    try:
        y_g, st_g, dx_g, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
        y_g2, st_g2, _, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
    finally:
        nv.call('segnb_tune', b'fprop_dma_cfg', -1)
    with on_emulator():
        y_e, st_e, dx_e, _, _, _ = _run_conv('cpu', 'bf16', full, w, b, x, dy)
    check(name + ' y', y_g, y_e, 'bf16')
    check(name + ' dx', dx_g, dx_e, 'bf16')
    np.testing.assert_allclose(st_g.numpy(), st_e.numpy(), rtol=2e-3, atol=2e-2 * float(st_e.abs().max()))
    assert float(y_g[..., Co:].abs().max()) == 0.0 if y_g.shape[-1] > Co else True
    assert torch.equal(y_g, y_g2)                                  # same bits run to run
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, b, padding=1)
    yr.backward(dy)
    check(name + ' y vs torch', y_g[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
    parts, off = [], 0
    for real, padded in segs:
        parts.append(dx_g[..., off:off + real])
        off += padded
    check(name + ' dx vs torch', torch.cat(parts, -1).permute(0, 3, 1, 2), xr.grad, 'bf16')


@pytest.mark.parametrize('shape', [(32, 56, 128, 128), (32, 14, 1536, 512), (32, 28, 256, 768), (16, 112, 192, 64),
                                   (32, 224, 32, 32), (32, 224, 96, 32), (32, 112, 32, 64), (32, 112, 64, 32),
                                   (32, 7, 1024, 1024), (32, 7, 512, 1024), (32, 224, 8, 32), (5, 37, 8, 24)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_fprop_dma_full_size_reproducible(shape):
    """bs=32 layer shapes of BASELINE.json configs[1]: many tiles per persistent block, tile seams, look-ahead reads in
    flight across the epilogue.  Bitwise equal outputs over repeated launches, and equal (up to accumulation order) to
    the register-staged kernels the pipeline replaces."""
    N, S, Ci, Co = shape
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(S + Ci)
    wt = (torch.randn(Co, Ci, 3, 3, generator=gen) * (2.0 / (Ci * 9)) ** 0.5).cuda()
    op = ConvOp(rt, wt, torch.zeros(Co, device='cuda'), [(Ci, Ci)], 1, 1, False, True)
    op.pack(S, S)
    xv = View.alloc(rt, N, S, S, op.Cip)
    xv.t.normal_()
    outs, sts = [], []
    for dma, reps in ((1, 4), (0, 1)):
        nv.call('segnb_tune', b'fprop_dma', dma)
        try:
            for rep in range(reps):
                yv = View.alloc(rt, N, S, S, op.Cop)
                stats = rt.zeros((16, 2, op.Cop), torch.float64)
                if rep % 2 == 0:
                    # cold caches (weights and input from HBM): the timing in which a fetch that a counted wait does
                    # not cover is still in flight when its LDS rows are read
                    flush = torch.empty(160 << 20, dtype=torch.float32, device='cuda').fill_(1.0)
                    del flush
                    torch.cuda.synchronize()
                op.fprop(xv, yv, stats)
                torch.cuda.synchronize()
                outs.append(yv.dense().clone())
                sts.append(stats.sum(0).cpu())
        finally:
            nv.call('segnb_tune', b'fprop_dma', 1)
    for k in range(1, 4):
        assert torch.equal(outs[0], outs[k]), 'launch %d differs from launch 0' % k
        np.testing.assert_allclose(sts[k].numpy(), sts[0].numpy(), rtol=1e-12)
    check('dma vs register-staged y', outs[0], outs[4], 'bf16')
    np.testing.assert_allclose(sts[0].numpy(), sts[4].numpy(), rtol=1e-3, atol=1e-3 * float(sts[4].abs().max()))


# every distinct (size, Ci, Co) of the timed configuration (lib/models/zf_unet.py:44-56)
FULL_SIZE_LAYERS = [(32, 224, 3, 32), (32, 224, 32, 32), (32, 112, 32, 64), (32, 112, 64, 64), (32, 56, 64, 128),
                    (32, 56, 128, 128), (32, 28, 128, 256), (32, 28, 256, 256), (32, 14, 256, 512), (32, 14, 512, 512),
                    (32, 7, 512, 1024), (32, 7, 1024, 1024), (32, 14, 1536, 512),
                    (32, 28, 768, 256), (32, 56, 384, 128), (32, 112, 192, 64), (32, 224, 96, 32)]


@pytest.mark.parametrize('wg_cu_pct', [0, 100], ids=['wg-default-cus', 'wg-all-cus'])
@pytest.mark.parametrize('shape', FULL_SIZE_LAYERS, ids=lambda s: 'x'.join(map(str, s)))
def test_conv_full_size_vs_torch(shape, wg_cu_pct):
    """The layer shapes of the TIMED configuration (BASELINE.json configs[1]: ZF_UNET 224x224 bs=32, bf16) through
    ConvOp exactly as the training step launches them -- forward, data gradient AND weight gradient, the latter with the
    default share of the CUs (multi-slab split, what the two-stream step runs) and with all CUs -- against
    F.conv2d autograd on the CPU (fp32 math on the same bf16-rounded operands): the oracle, not another HIP kernel.
    (VERDICT r1 weak #3.)"""
    N, S, Ci, Co = shape
    first = Ci == 3
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(S * 1000 + Ci + Co)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) * (2.0 / (Ci * 9)) ** 0.5).bfloat16().float()
    b = (torch.randn(Co, generator=gen) * 0.1)
    x = torch.randn(N, Ci, S, S, generator=gen).bfloat16().float()
    dy = torch.randn(N, Co, S, S, generator=gen).bfloat16().float()
    Cip = cp.pad8(Ci)
    nv.call('segnb_tune', b'wg_cu_pct', wg_cu_pct)
    try:
        op = ConvOp(rt, w.cuda(), b.cuda(), [(Ci, Cip)], 1, 1, False, need_dgrad=not first)
        op.pack(S, S)
        xv = View.alloc(rt, N, S, S, op.Cip)
        xv.dense()[..., :Ci] = x.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
        yv = View.alloc(rt, N, S, S, op.Cop)
        stats = rt.zeros((16, 2, op.Cop), torch.float64)
        op.fprop(xv, yv, stats)
        dyv = View.alloc(rt, N, S, S, op.Cop)
        dyv.dense()[..., :Co] = dy.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
        gw = torch.zeros_like(op.weight)
        op.wgrad(xv, dyv, gw)
        gw2 = torch.zeros_like(op.weight)
        op.wgrad(xv, dyv, gw2)
        dxv = None
        if not first:
            dxv = View.alloc(rt, N, S, S, op.Cip)
            op.dgrad(dyv, dxv)
        torch.cuda.synchronize()
    finally:
        nv.call('segnb_tune', b'wg_cu_pct', 0)
    xr = x.clone().requires_grad_(not first)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=1)
    yr.backward(dy)
    name = 'x'.join(map(str, shape))
    y_g = yv.dense().float().cpu()
    check(name + ' y vs torch', y_g[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
    assert op.Cop == Co or float(y_g[..., Co:].abs().max()) == 0.0
    # BatchNorm statistics of the epilogue = sums of the STORED (bf16) outputs
    st = stats.sum(0).cpu()
    ys = yr.detach().bfloat16().double()
    np.testing.assert_allclose(st[0, :Co].numpy(), ys.sum((0, 2, 3)).numpy(), rtol=1e-3,
                               atol=2e-2 * float(ys.abs().max()) * (N * S * S) ** 0.5)
    np.testing.assert_allclose(st[1, :Co].numpy(), (ys * ys).sum((0, 2, 3)).numpy(), rtol=2e-3)
    # weight gradient: fp32 accumulation over N*S*S pixels on both sides
    check(name + ' dw vs torch', gw.cpu(), wr.grad, 'f32')
    assert torch.equal(gw, gw2), 'weight gradient differs between two launches (slab reduction order)'
    if not first:
        check(name + ' dx vs torch', dxv.dense().float().cpu()[..., :Ci].permute(0, 3, 1, 2), xr.grad, 'bf16')


DIRECT_DW_SHAPES = [  # (N, H, W, input segments [(real, padded)], Co): what the launch is served by
    (4, 32, 32, [(32, 32)], 32),          # thin 32 x 32 tiles, many slabs
    (2, 64, 64, [(3, 8)], 32),            # first layer: rolling kernel, 3 real of 8 input channels
    (2, 56, 56, [(64, 64)], 128),         # wave-specialised 64 x 64 tiles, pixel split
    (8, 14, 14, [(256, 256)], 192),       # flattened 14 x 14 tiles: one slab, written from the accumulators
    (8, 7, 7, [(128, 128)], 64),          # flattened 7 x 7 tiles
    (2, 28, 28, [(192, 192)], 40),        # ragged output channels (40 of a 64-channel tile), 16-column tiles
    (2, 24, 40, [(24, 24)], 24),          # ragged everything on thin tiles
    (1, 16, 16, [(12, 16), (6, 8)], 16),  # padded concat segments: NOT in place -- stays on the workspace + unpack path
]


@pytest.mark.parametrize('accumulate', [False, True], ids=['fresh', 'accumulate'])
@pytest.mark.parametrize('shape', DIRECT_DW_SHAPES, ids=lambda s: 'x'.join(str(v) for v in (s[0], s[1], s[2], sum(r for r, _ in s[3]), s[4])))
def test_weight_gradient_delivered_into_the_parameter_gradient(shape, accumulate):
    """segnb_wgrad_target: the weight-gradient launches add their result to the parameter's own fp32 gradient
    ([Co][Ci][3][3], torch_train.py:188's .grad) -- slab sum, transposition and accumulation in one pass, or straight from the
    accumulators of a single-slab launch -- instead of leaving it in the packed workspace for segnb_unpack_wgrad_multi.
    Oracle: F.conv2d autograd on the CPU (fp32 on the same bf16 operands); also == the workspace + unpack path up to the
    order of the slab sum, bitwise equal run to run, on top of what the gradient already held."""
    N, H, W, segs, Co = shape
    Ci = sum(r for r, _ in segs)
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 100 + Ci + Co)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) * 0.1).bfloat16().float()
    x = torch.randn(N, Ci, H, W, generator=gen).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=gen).bfloat16().float()
    base = torch.randn(Co, Ci, 3, 3, generator=gen) if accumulate else torch.zeros(Co, Ci, 3, 3)

    def run(direct):
        op = ConvOp(rt, w.cuda(), None, segs, 1, 1, False, need_dgrad=False)
        op.direct_dw = direct
        xv = View.alloc(rt, N, H, W, op.Cip)
        off, roff = 0, 0
        for real, padded in segs:
            xv.dense()[..., off:off + real] = x[:, roff:roff + real].permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
            off, roff = off + padded, roff + real
        dyv = View.alloc(rt, N, H, W, op.Cop)
        dyv.dense()[..., :Co] = dy.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
        outs = []
        for _ in range(2):
            gw = base.clone().cuda()
            op.wgrad(xv, dyv, gw)
            torch.cuda.synchronize()
            outs.append(gw)
        return op, outs

    op_d, (g1, g2) = run(True)
    in_place = len(segs) == 1
    assert op_d.direct_ok() == in_place
    assert not in_place or op_d.unpack_jobs(H, W, g1) == []
    assert torch.equal(g1, g2), 'not reproducible run to run'
    _, (gu, _) = run(False)
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(dy)
    name = 'direct dw ' + 'x'.join(map(str, (N, H, W, Ci, Co)))
    check(name + ' vs torch', (g1.cpu() - base), wr.grad, 'f32')
    scale = float(wr.grad.abs().max())
    assert float((g1 - gu).abs().max()) <= 2e-6 * scale * max(1.0, (N * H * W) ** 0.5 / 16), 'differs from the workspace + unpack path'


@pytest.mark.parametrize('shape', [(32, 224, 224, 3, 32, 1), (2, 40, 56, 3, 32, 2), (3, 33, 47, 8, 24, 1), (2, 64, 64, 32, 32, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_wgrad_with_recomputed_bn_apply(shape, monkeypatch):
    """segnb_conv_wgrad_bnapply: the weight gradient of a layer whose dy operand -- the BatchNorm-backward apply of
    (g, y), lib/modules/abn/functions.py:118 -- is recomputed while the tiles are staged (first layer of the network: the
    apply pass disappears) == segnb_bn_bwd_apply_direct followed by segnb_conv_wgrad, bit for bit; incl. the first layer
    of the timed configuration (bs=32 224x224, 3 -> 32)."""
    N, H, W, Ci, Co, act = shape
    monkeypatch.setenv('SEGNB_WGRAD_BNAPPLY', '1')     # (default: only the first layer's rolling kernel -- 8 input channels)
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 3 + Co)
    w = torch.randn(Co, Ci, 3, 3, generator=gen).cuda()
    Cip = cp.pad8(Ci)
    op = ConvOp(rt, w, None, [(Ci, Cip)], 1, 1, False, need_dgrad=False)
    op.direct_dw = False                # (the two results are compared in the packed workspace)
    xv = View.alloc(rt, N, H, W, Cip)
    xv.dense()[..., :Ci].normal_()
    yv = View.alloc(rt, N, H, W, op.Cop)
    yv.t.normal_()
    gv = View.alloc(rt, N, H, W, op.Cop)
    gv.t.normal_()
    Cp = op.Cop
    coef = torch.stack([0.5 + torch.rand(Cp, generator=gen), 0.3 * torch.randn(Cp, generator=gen),
                        0.2 * torch.randn(Cp, generator=gen), 0.5 + torch.rand(Cp, generator=gen)]).cuda()
    bcoef = torch.stack([0.5 + torch.rand(Cp, generator=gen), 0.1 * torch.randn(Cp, generator=gen),
                         0.1 * torch.randn(Cp, generator=gen)]).cuda()
    assert op.wgrad_bnapply_ok(xv, yv)
    monkeypatch.delenv('SEGNB_WGRAD_BNAPPLY')
    assert op.wgrad_bnapply_ok(xv, yv) == (Cip == 8 and W >= 32)
    monkeypatch.setenv('SEGNB_WGRAD_BNAPPLY', '0')
    assert not op.wgrad_bnapply_ok(xv, yv)
    monkeypatch.setenv('SEGNB_WGRAD_BNAPPLY', '1')
    slope = 0.01
    dz = View.alloc(rt, N, H, W, Cp)
    nv.call('segnb_bn_bwd_apply_direct', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), act, slope,
            gv.ptr, gv.ld, dz.ptr, dz.ld, None, Co, rt.stream)
    p = op.plan(H, W)
    op.wgrad(xv, dz, torch.zeros_like(w), unpack=False)
    ref = p['dwp'][0][0].clone()
    p['dwp'][0].zero_()
    op.wgrad_bnapply(xv, gv, yv, coef, bcoef, act, slope)
    got = p['dwp'][0][0].clone()
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 0
    assert torch.equal(got, ref), float((got - ref).abs().max())


# (N, high-resolution size, channels of the upsampled tensor, skip channels, output channels): the five decoder levels of
# the timed configuration, a ragged toy, and sizes that take the general kernels (odd, non-multiples of 64)
UPCAT_SHAPES = [(32, 14, 1024, 512, 512), (32, 28, 512, 256, 256), (32, 56, 256, 128, 128), (32, 112, 128, 64, 64),
                (32, 224, 64, 32, 32), (3, 12, 24, 12, 20), (2, 20, 64, 64, 64), (2, 36, 128, 64, 64),
                (3, 36, 64, 32, 32), (2, 20, 16, 24, 32)]


@pytest.mark.parametrize('shape', UPCAT_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_upcat_segmented_backward_vs_torch(shape):
    """UpCatConvOp (the decoder blocks' first convolution, lib/models/zf_unet.py:78-90: cat([Upsample x2(u), skip]) ->
    conv3x3): data gradient of the skip segment, LOW-resolution gradient of u and the weight gradient of both segments --
    the upsampled one through the ConvTranspose2d(4, 2, 1) identity on the low-resolution tensor with masked pack /
    unpack jobs -- against torch autograd of F.conv2d(cat(F.interpolate(nearest), skip)) on the CPU, at the five decoder
    shapes of the timed configuration (bs=32) and at ragged sizes.  Weight gradients land in the reference's 3x3 layout."""
    N, S, Cu, Cs, Co = shape
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(S * 131 + Cu + Co)
    Ci = Cu + Cs
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) * (2.0 / (Ci * 9)) ** 0.5)
    u = torch.randn(N, Cu, S // 2, S // 2, generator=gen).bfloat16().float()
    sk = torch.randn(N, Cs, S, S, generator=gen).bfloat16().float()
    dy = torch.randn(N, Co, S, S, generator=gen).bfloat16().float()
    Cup, Csp = cp.pad8(Cu), cp.pad8(Cs)
    wd = w.cuda()
    op = UpCatConvOp(rt, wd, None, [(Cu, Cup), (Cs, Csp)], need_dgrad=True)
    op.segment_wgrad = True                            # (instance overrides: both segmented paths are exercised here,
    op.force_segmented = True                          #  on the general kernels where no fast one serves the shape)
    op.segment_fwd = True
    # every packed matrix of the three ops (the plan packs only what its mode at this size reads: UpCatConvOp.pack_jobs)
    PackTable(rt, op.full.pack_jobs(S, S) + op.skip.pack_jobs(S, S) + op.up.pack_jobs(S // 2, S // 2),
              'segnb_pack_weight_multi', 'segnb_pack_weight').run()
    cat = View.alloc(rt, N, S, S, Cup + Csp)
    cat.dense()[..., :Cu] = F.interpolate(u, scale_factor=2, mode='nearest').permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
    cat.dense()[..., Cup:Cup + Cs] = sk.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
    uv = View.alloc(rt, N, S // 2, S // 2, Cup)
    uv.dense()[..., :Cu] = u.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
    duv = View.alloc(rt, N, S // 2, S // 2, Cup)
    duv.t.fill_(7.0)                                   # every element must be overwritten
    dcat = View.alloc(rt, N, S, S, Cup + Csp)
    dyv = View.alloc(rt, N, S, S, op.Cop)
    dyv.dense()[..., :Co] = dy.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
    yv = View.alloc(rt, N, S, S, op.Cop)
    op.fprop(cat, yv)                                  # (no low-resolution tensor bound yet: the one 9-tap launch)
    op.bind_up(uv, duv)
    cat_noup = View.alloc(rt, N, S, S, Cup + Csp)              # the upsampled copy is NOT read below: leave it as garbage
    cat_noup.t.fill_(5.0)
    cat_noup.dense()[..., Cup:] = cat.dense()[..., Cup:]
    # (a) VIRTUAL CONCAT: the 9-tap forward / weight gradient reading the upsampled segment from the low-resolution tensor
    yvirt, vstats, gwv = None, None, None
    op.virtual_concat, op.segment_wgrad = True, False
    op._seg.clear()
    if op.virtual(N, S, S):
        yvirt = View.alloc(rt, N, S, S, op.Cop)
        yvirt.t.fill_(3.0)
        vstats = rt.zeros((16, 2, op.Cop), torch.float64)
        op.fprop(cat_noup, yvirt, vstats)
        gwv = torch.zeros_like(wd)
        op.wgrad(cat_noup, dyv, gwv, unpack=False)
        PackTable(rt, op.full.unpack_jobs(S, S, gwv), 'segnb_unpack_wgrad_multi', 'segnb_unpack_wgrad').run()
    # (b) forward BY SEGMENT: the skip segment's 9-tap launch, then the upsampled segment added on the low-resolution tensor
    op.virtual_concat, op.segment_wgrad = False, True
    op._seg.clear()
    yseg, stats = None, None
    if op.fwd_segmented(N, S, S, op.Cop):
        yseg = View.alloc(rt, N, S, S, op.Cop)
        yseg.t.fill_(3.0)
        stats = rt.zeros((16, 2, op.Cop), torch.float64)
        op.fprop(cat_noup, yseg, stats)
    op.dgrad(dyv, dcat)
    # (c) the PLAIN data gradient with the Upsample backward fused into its store pass (the thin 224x224 level)
    op.force_segmented = False
    op._seg.clear()
    dcat_s, duv_s = None, None
    if op.upsum(N, S, S):
        dcat_s = View.alloc(rt, N, S, S, Cup + Csp)
        dcat_s.t.fill_(9.0)
        duv_s = View.alloc(rt, N, S // 2, S // 2, Cup)
        duv_s.t.fill_(7.0)
        op.bind_up(uv, duv_s)
        assert op.writes_du(N, S, S) and not op.segmented(N, S, S)
        op.dgrad(dyv, dcat_s)
        op.bind_up(uv, duv)
    assert dcat_s is not None or Co > 32 or (Co % 32), 'fused upsample backward not served at a thin decoder shape'
    op.force_segmented = True
    op._seg.clear()
    gw = torch.zeros_like(wd)
    op.wgrad(cat, dyv, gw, unpack=False)
    unp = PackTable(rt, op.unpack_jobs(S, S, gw), 'segnb_unpack_wgrad_multi', 'segnb_unpack_wgrad')
    unp.run()
    gw1 = gw.clone()
    gw.zero_()
    op.wgrad(cat, dyv, gw, unpack=False)               # second round: consumed workspaces, same bits
    unp.run()
    torch.cuda.synchronize()
    # reference on the CPU: fp32 math on the operands the kernels saw (weights rounded to bf16 tap by tap for the forward
    # and the skip segment; the up segment's packed taps are rounded AFTER the fp32 sum -- within bf16 tolerance of it)
    ur = u.clone().requires_grad_(True)
    skr = sk.clone().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    yr = F.conv2d(torch.cat([F.interpolate(ur, scale_factor=2, mode='nearest'), skr], 1), wr, None, padding=1)
    yr.backward(dy)
    name = 'x'.join(map(str, shape))
    check(name + ' y', yv.dense().float().cpu()[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
    for tag, yvw, stw in (('by segment', yseg, stats), ('virtual concat', yvirt, vstats)):
        if yvw is None:
            continue
        ys = yvw.dense().float().cpu()
        check(name + ' y ' + tag, ys[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
        st = stw.sum(0).cpu()
        yd = ys[..., :Co].double()
        np.testing.assert_allclose(st[0, :Co].numpy(), yd.sum((0, 1, 2)).numpy(), rtol=1e-6, atol=1e-3)
        np.testing.assert_allclose(st[1, :Co].numpy(), (yd * yd).sum((0, 1, 2)).numpy(), rtol=1e-6)
    if yvirt is not None:
        # the same launches on the same operands as the 9-tap convolution over the materialised concat: the same bits
        assert torch.equal(yvirt.dense()[..., :Co], yv.dense()[..., :Co])
        check(name + ' dw virtual concat', gwv.cpu(), wr.grad, 'f32')
    assert yvirt is not None or S // 2 < 12 or (Cu % 32) or (Cs % 32), 'virtual concat not served at a decoder shape' 
    check(name + ' d skip', dcat.dense().float().cpu()[..., Cup:Cup + Cs].permute(0, 3, 1, 2), skr.grad, 'bf16')
    check(name + ' d u (low resolution)', duv.dense().float().cpu()[..., :Cu].permute(0, 3, 1, 2), ur.grad, 'bf16')
    assert Cup == Cu or float(duv.dense()[..., Cu:].abs().max()) == 0.0
    check(name + ' dw', gw.cpu(), wr.grad, 'f32')
    check(name + ' dw, second launch', gw1.cpu(), wr.grad, 'f32')
    if dcat_s is not None:
        check(name + ' d skip (fused upsample backward)', dcat_s.dense().float().cpu()[..., Cup:Cup + Cs].permute(0, 3, 1, 2),
              skr.grad, 'bf16')
        check(name + ' d u (fused upsample backward)', duv_s.dense().float().cpu()[..., :Cu].permute(0, 3, 1, 2), ur.grad, 'bf16')
        assert float((dcat_s.dense()[..., :Cup] - 9.0).abs().max()) == 0.0, 'the high-resolution slice is not to be written'
        # = the plain data gradient's stored slice, summed 2 x 2 in fp32 and rounded once
        dfull = View.alloc(rt, N, S, S, Cup + Csp)
        op.full.dgrad(dyv, dfull)
        f = dfull.dense()[..., :Cup].float()
        ref = (((f[:, 0::2, 0::2] + f[:, 0::2, 1::2]) + f[:, 1::2, 0::2]) + f[:, 1::2, 1::2]).bfloat16()
        assert torch.equal(duv_s.dense(), ref)
        assert torch.equal(dcat_s.dense()[..., Cup:], dfull.dense()[..., Cup:])


@pytest.mark.parametrize('shape', [(3, 40, 56, 32, 32, 1), (2, 33, 47, 32, 32, 2), (32, 224, 224, 32, 32, 1), (2, 24, 40, 32, 64, 1),
                                   # a dense layer's data gradient (tiramisu.py:9-20: growth 16 -> the concat prefix) on the
                                   # general gather kernel's store pass: 8 x 8 maps, ragged channel tiles, FCDenseNet103's sizes
                                   (2, 24, 40, 112, 16, 1), (3, 9, 11, 48, 16, 2), (8, 8, 8, 1072, 16, 1), (8, 16, 16, 656, 16, 1),
                                   (8, 128, 128, 272, 16, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_dgrad_with_fused_bn_reduce(shape):
    """segnb_conv_fprop_bnreduce: the data-gradient launch whose epilogue does the BatchNorm-backward reduction of the
    layer that produced its input (the edz_eydz phase of lib/modules/abn/functions.py:112 folded into
    the producer of dz).  dx is bit-identical to the plain data gradient; the sums equal segnb_bn_act_bwd_reduce on that
    dx (same per-element arithmetic, another summation order) and the emulator's."""
    N, H, W, C1, C2, act = shape           # layer 1: ? -> C1 (BatchNorm, act);  layer 2: C1 -> C2
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 7 + C2)
    w2 = (torch.randn(C2, C1, 3, 3, generator=gen) * (2.0 / (C1 * 9)) ** 0.5).cuda()
    op = ConvOp(rt, w2, None, [(C1, C1)], 1, 1, False, True)
    op.pack(H, W)
    dyv = View.alloc(rt, N, H, W, op.Cop)
    dyv.t.normal_(generator=None)
    y1 = View.alloc(rt, N, H, W, C1)
    y1.t.normal_()
    coef = torch.stack([0.5 + torch.rand(C1, generator=gen), 0.3 * torch.randn(C1, generator=gen),
                        0.2 * torch.randn(C1, generator=gen), 0.5 + torch.rand(C1, generator=gen)]).cuda().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    dx_plain = View.alloc(rt, N, H, W, C1)
    op.dgrad(dyv, dx_plain)
    sums_ref = rt.zeros((16, 2, C1), torch.float64)
    nv.call('segnb_bn_act_bwd_reduce', rt.code, y1.ptr, y1.ld, N, H, W, C1, nv.ptr(coef), act, 0.01, None, dx_plain.ptr,
            dx_plain.ld, None, 0, None, 0, None, 0, nv.ptr(sums_ref), None, 0, st)
    assert op.dgrad_bnreduce_ok(dyv, dx_plain)
    dx_f = View.alloc(rt, N, H, W, C1)
    sums_f = rt.zeros((16, 2, C1), torch.float64)
    op.dgrad(dyv, dx_f, bn_reduce=(y1, coef, sums_f, act, 0.01))
    sums_f2 = rt.zeros((16, 2, C1), torch.float64)
    op.dgrad(dyv, dx_f, bn_reduce=(y1, coef, sums_f2, act, 0.01))
    torch.cuda.synchronize()
    if C2 == 32 or C2 == 16:
        assert torch.equal(dx_f.t, dx_plain.t)          # (the same kernel with and without the epilogue)
    else:
        # 64 -> 32: the plain data gradient runs on conv_roll_kernel with the K split over two waves, the fused one on
        # conv_fprop_rw_kernel -- another summation order of the same products
        check('dx fused vs plain', dx_f.t, dx_plain.t, 'bf16')
    a, b = sums_f.sum(0).cpu().numpy(), sums_ref.sum(0).cpu().numpy()
    scale = np.abs(b).max(axis=1, keepdims=True) + 1e-30
    tight = C2 == 32 or C2 == 16    # (sums_ref is taken on dx_plain: the same values bit for bit only when the kernels are the same)
    assert np.abs(a - b).max() <= (2e-5 if tight else 2e-3) * float(np.abs(b).max()) + 1e-6 * (N * H * W) ** 0.5, np.abs(a - b).max()
    np.testing.assert_allclose(a / scale, b / scale, atol=1e-4 if tight else 3e-3)
    # fixed summation order inside a block, fp64 across blocks: run-to-run equal to fp64 rounding
    np.testing.assert_allclose(sums_f2.sum(0).cpu().numpy(), a, rtol=1e-12, atol=1e-9)
    if N * H * W <= 20000:
        with on_emulator():
            sums_e = torch.zeros((16, 2, C1), dtype=torch.float64)
            ye, ce, de = y1.t.cpu(), coef.cpu(), dx_plain.t.cpu()
            EMU.segnb_bn_act_bwd_reduce(nv.BF16, ye.data_ptr(), C1, N, H, W, C1, ce.data_ptr(), act, 0.01, None,
                                        de.data_ptr(), C1, None, 0, None, 0, None, 0, sums_e.data_ptr(), None, 0, 0)
        e = sums_e.sum(0).numpy()
        np.testing.assert_allclose(a / scale, e / scale, atol=2e-4)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('with_res', [False, True])
@pytest.mark.parametrize('shape', [(2, 17, 23, 40), (1, 64, 64, 64), (16, 128, 128, 64)], ids=lambda s: 'x'.join(map(str, s)))
def test_bn_act_bwd_reduce_two_sources(shape, with_res, dtype):
    """segnb_bn_act_bwd_reduce_add(g1, g2) against the oracle -- torch's activation backward of round(g1 + g2) on the CPU, float64
    sums -- and == segnb_add(g1, g2) followed by segnb_bn_act_bwd_reduce: dz bit for bit, the sums to fp64 rounding of another
    block order (the identity branches of linknet.py:41-62: a tensor with two consumers)."""
    N, H, W, C = shape
    rt = Runtime('cuda', dtype)
    y, g1, g2, r = (View.alloc(rt, N, H, W, C) for _ in range(4))
    for v in (y, g1, g2, r):
        v.t.normal_()
    gen = torch.Generator().manual_seed(C)
    coef = torch.stack([0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen),
                        0.2 * torch.randn(C, generator=gen), 0.5 + torch.rand(C, generator=gen)]).cuda().contiguous()
    st = rt.stream
    gs = View.alloc(rt, N, H, W, C)
    nv.call('segnb_add', rt.code, g1.ptr, g1.ld, g2.ptr, g2.ld, gs.ptr, gs.ld, N, H, W, C, st)
    dz_ref, dz = View.alloc(rt, N, H, W, C), View.alloc(rt, N, H, W, C)
    s_ref, s_f = rt.zeros((16, 2, C), torch.float64), rt.zeros((16, 2, C), torch.float64)
    rp, rl = (r.ptr, r.ld) if with_res else (None, 0)
    nv.call('segnb_bn_act_bwd_reduce', rt.code, y.ptr, y.ld, N, H, W, C, nv.ptr(coef), nv.ACT_LEAKY, 0.01, None, gs.ptr, gs.ld,
            None, 0, None, 0, dz_ref.ptr, dz_ref.ld, nv.ptr(s_ref), rp, rl, st)
    nv.call('segnb_bn_act_bwd_reduce_add', rt.code, y.ptr, y.ld, N, H, W, C, nv.ptr(coef), nv.ACT_LEAKY, 0.01, None, g1.ptr, g1.ld,
            g2.ptr, g2.ld, dz.ptr, dz.ld, nv.ptr(s_f), rp, rl, st)
    torch.cuda.synchronize()
    # oracle: torch's own activation backward on g = round(g1 + g2) (what torch.add of two bf16 / fp32 gradients holds), float64 sums
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g_t = (g1.dense().float().cpu() + g2.dense().float().cpu()).to(tdt)
    dz_t, s1_t, s2_t = torch_bn_act_bwd(y.dense(), g_t, coef, nv.ACT_LEAKY, 0.01, dtype, r.dense() if with_res else None)
    check('dz vs torch', dz.dense(), dz_t, dtype)
    a = s_f.sum(0).cpu().numpy()
    np.testing.assert_allclose(a[0], s1_t.numpy(), rtol=1e-4, atol=1e-4 * float(s1_t.abs().max()) + 1e-3)
    np.testing.assert_allclose(a[1], s2_t.numpy(), rtol=1e-4, atol=1e-4 * float(s2_t.abs().max()) + 1e-3)
    # sibling: the two separate launches it replaces, bit for bit
    assert torch.equal(dz.t, dz_ref.t)
    b = s_ref.sum(0).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9 * float(np.abs(b).max()))
    if N * H * W <= 20000:
        with on_emulator():
            rc = Runtime('cpu', dtype)
            ye, g1e, g2e, re_, dze = (View.alloc(rc, N, H, W, C) for _ in range(5))
            for d, srcv in ((ye, y), (g1e, g1), (g2e, g2), (re_, r)):
                d.t.copy_(srcv.t.cpu())
            se, ce = rc.zeros((16, 2, C), torch.float64), coef.cpu()
            nv.call('segnb_bn_act_bwd_reduce_add', rc.code, ye.ptr, ye.ld, N, H, W, C, nv.ptr(ce), nv.ACT_LEAKY, 0.01, None, g1e.ptr,
                    g1e.ld, g2e.ptr, g2e.ld, dze.ptr, dze.ld, nv.ptr(se), re_.ptr if with_res else None, re_.ld if with_res else 0,
                    rc.stream)
        check('dz vs emulator', dz.t, dze.t, dtype)


@pytest.mark.parametrize('shape', [(2, 40, 56, 32, 32, nv.ACT_RELU, 1), (3, 33, 47, 24, 32, nv.ACT_LEAKY, 1),
                                   (2, 38, 45, 32, 32, nv.ACT_LEAKY, 0),       # linknet.py:60 finalconv2 = Conv2d(32, 32, 3): valid window
                                   (16, 511, 511, 32, 32, nv.ACT_LEAKY, 0),
                                   # conv_fprop_ws_kernel's MASK instantiation (unet16.py:73-108: VGG-style conv + ReLU stacks)
                                   (2, 40, 56, 64, 64, nv.ACT_RELU, 1), (3, 33, 47, 72, 128, nv.ACT_LEAKY, 1),
                                   (2, 38, 45, 64, 64, nv.ACT_LEAKY, 0), (2, 20, 24, 40, 192, nv.ACT_RELU, 1),
                                   (4, 128, 160, 256, 64, nv.ACT_RELU, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_dgrad_with_act_mask(shape):
    """segnb_conv_fprop_bnreduce with coef NULL: the data gradient whose store pass applies the activation mask of the conv +
    activation (no BatchNorm) that produced its input -- dx == segnb_bn_act_bwd_reduce(plain dx, activated tensor) bit for bit,
    the sums equal that pass's (another summation order) and the emulator's."""
    N, H, W, C1, C2, act, pad = shape        # layer 1: ? -> C1 channels, activation;  layer 2: C1 -> C2, 3 x 3, padding `pad`
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 7 + C2 + pad)
    w2 = (torch.randn(C2, C1, 3, 3, generator=gen) * (2.0 / (C1 * 9)) ** 0.5).cuda()
    op = ConvOp(rt, w2, None, [(C1, cp.pad8(C1))], 1, pad, False, True)
    op.pack(H, W)
    Ho, Wo = op.out_hw(H, W)
    C1p = cp.pad8(C1)
    dyv = View.alloc(rt, N, Ho, Wo, op.Cop)
    dyv.t.normal_()
    a1 = View.alloc(rt, N, H, W, C1p)          # the ACTIVATED output of layer 1
    a1.t.normal_()
    if act == nv.ACT_RELU:
        a1.t.clamp_(min=0)
    st = torch.cuda.current_stream().cuda_stream
    dx_plain = View.alloc(rt, N, H, W, C1p)
    op.dgrad(dyv, dx_plain)
    dz_ref = View.alloc(rt, N, H, W, C1p)
    sums_ref = rt.zeros((16, 2, C1p), torch.float64)
    nv.call('segnb_bn_act_bwd_reduce', rt.code, a1.ptr, a1.ld, N, H, W, C1p, None, act, 0.01, None, dx_plain.ptr, dx_plain.ld,
            None, 0, None, 0, dz_ref.ptr, dz_ref.ld, nv.ptr(sums_ref), None, 0, st)
    assert op.dgrad_actmask_ok(dyv, dx_plain)
    dz_f = View.alloc(rt, N, H, W, C1p)
    sums_f = rt.zeros((16, 2, C1p), torch.float64)
    op.dgrad(dyv, dz_f, bn_reduce=(a1, None, sums_f, act, 0.01))
    torch.cuda.synchronize()
    assert torch.equal(dz_f.t, dz_ref.t)
    a, b = sums_f.sum(0).cpu().numpy(), sums_ref.sum(0).cpu().numpy()
    assert np.abs(a[1]).max() == 0.0
    assert np.abs(a[0] - b[0]).max() <= 2e-5 * float(np.abs(b[0]).max()) + 1e-6 * (N * H * W) ** 0.5, np.abs(a[0] - b[0]).max()
    if N * H * W <= 20000:
        def run_emu():
            r = Runtime('cpu', 'bf16')
            ope = ConvOp(r, w2.cpu(), None, [(C1, C1p)], 1, pad, False, True)
            ope.pack(H, W)
            dye, ae, dze = View.alloc(r, N, Ho, Wo, ope.Cop), View.alloc(r, N, H, W, C1p), View.alloc(r, N, H, W, C1p)
            dye.t.copy_(dyv.t.cpu())
            ae.t.copy_(a1.t.cpu())
            se = r.zeros((16, 2, C1p), torch.float64)
            assert ope.dgrad_actmask_ok(dye, dze)
            ope.dgrad(dye, dze, bn_reduce=(ae, None, se, act, 0.01))
            return dze.t.float(), se.sum(0).numpy()
        with on_emulator():
            dz_e, s_e = run_emu()
        check('dz vs emulator', dz_f.t, dz_e, 'bf16')
        scale = np.abs(b[0]).max() + 1e-30
        np.testing.assert_allclose(a[0] / scale, s_e[0] / scale, atol=2e-3)


@pytest.mark.parametrize('shape', [(8, 16, 16, 656, 16), (8, 8, 8, 1072, 16), (2, 64, 64, 112, 16), (2, 12, 14, 128, 16), (3, 40, 72, 160, 12),
                                   (4, 32, 32, 256, 16)], ids=lambda s: 'x'.join(map(str, s)))
def test_conv_fprop_with_dropout_and_slice_statistics(shape):
    """segnb_conv_fprop_drop (a dense layer of tiramisu.py:9-20: conv(C -> 16) -> Dropout2d, written into its slice of the concat buffer,
    the slice's statistics into the buffer's table) == segnb_conv_fprop + segnb_bn_act_fwd_stats(coef NULL, ACT_NONE, dropmul): the same
    bits in the slice, the same statistics up to summation order; and the emulator's."""
    N, H, W, C1, C2 = shape
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 13 + C1)
    w = (torch.randn(C2, C1, 3, 3, generator=gen) * (2.0 / (C1 * 9)) ** 0.5).cuda()
    b = (0.1 * torch.randn(C2, generator=gen)).cuda()
    C1p, C2p = cp.pad8(C1), cp.pad8(C2)
    op = ConvOp(rt, w, b, [(C1, C1p)], 1, 1, False, True)
    op.pack(H, W)
    xv = View.alloc(rt, N, H, W, C1p)
    xv.t.normal_()
    LD, OFF = C1p + C2p + 24, C1p                       # the slice sits behind C1p channels of a wider buffer
    cat_ref, cat_f = (View.alloc(rt, N, H, W, LD) for _ in range(2))
    drop = (torch.rand(N, C2p, generator=gen) > 0.2).float().mul_(1.25).cuda()
    table_ref, table_f = (rt.zeros((16, 2, LD), torch.float64) for _ in range(2))
    y = View.alloc(rt, N, H, W, C2p)
    st = rt.stream
    assert op.drop_epilogue_ok(N, H, W, LD), 'shape not served'
    op.fprop(xv, y, None)
    sl = cat_ref.slice(OFF, C2p)
    nv.call('segnb_bn_act_fwd_stats', rt.code, y.ptr, y.ld, N, H, W, C2p, None, nv.ACT_NONE, 0.0, nv.ptr(drop), sl.ptr, sl.ld,
            nv.ptr(table_ref, OFF), LD, st)
    op.fprop_drop(xv, cat_f.slice(OFF, C2p), drop, (table_f, OFF, LD))
    torch.cuda.synchronize()
    assert torch.equal(cat_f.t, cat_ref.t)
    a, r = table_f.sum(0).cpu().numpy(), table_ref.sum(0).cpu().numpy()
    assert np.abs(a[:, :OFF]).max() == 0.0 and np.abs(a[:, OFF + C2p:]).max() == 0.0          # nothing outside the slice's columns
    np.testing.assert_allclose(a, r, rtol=1e-5, atol=1e-5 * float(np.abs(r).max()))
    if N * H * W <= 20000:
        def run_emu():
            rc = Runtime('cpu', 'bf16')
            ope = ConvOp(rc, w.cpu(), b.cpu(), [(C1, C1p)], 1, 1, False, True)
            ope.pack(H, W)
            xe, ce = View.alloc(rc, N, H, W, C1p), View.alloc(rc, N, H, W, LD)
            xe.t.copy_(xv.t.cpu())
            te = rc.zeros((16, 2, LD), torch.float64)
            assert ope.drop_epilogue_ok(N, H, W, LD)
            ope.fprop_drop(xe, ce.slice(OFF, C2p), drop.cpu(), (te, OFF, LD))
            return ce.t.float(), te.sum(0).numpy()
        with on_emulator():
            ce, te = run_emu()
        check('slice vs emulator', cat_f.t, ce, 'bf16')
        np.testing.assert_allclose(a, te, rtol=2e-2, atol=2e-2 * float(np.abs(te).max()))


@pytest.mark.parametrize('accumulate', [0, 1])
@pytest.mark.parametrize('shape', [(2, 24, 40, 112, 16, 1), (3, 9, 11, 48, 16, 2), (8, 8, 8, 1072, 16, 1), (8, 16, 16, 656, 16, 1),
                                   (2, 17, 23, 72, 12, 1), (8, 128, 128, 272, 16, 1)], ids=lambda s: 'x'.join(map(str, s)))
def test_conv_dgrad_never_stored(shape, accumulate):
    """segnb_conv_fprop_bnsums + segnb_conv_fprop_bnapply (a dense layer's 16 -> prefix data gradient as two launches that never
    store it, tiramisu.py:9-20) against segnb_conv_fprop_bnreduce + segnb_bn_bwd_apply_fused_direct(_acc): the same sums, the same
    dx up to the rare element whose z or product rounds the other way in the other kernel, the same dgamma / dbeta / bcoef."""
    N, H, W, C1, C2, act = shape           # layer 1: BatchNorm(C1) + act (pre-activation);  layer 2: conv C1 -> C2
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 7 + C1)
    w2 = (torch.randn(C2, C1, 3, 3, generator=gen) * (2.0 / (C1 * 9)) ** 0.5).cuda()
    op = ConvOp(rt, w2, None, [(C1, C1)], 1, 1, False, True)
    op.pack(H, W)
    dyv = View.alloc(rt, N, H, W, op.Cop)
    dyv.t.normal_()
    y1 = View.alloc(rt, N, H, W, C1)
    y1.t.normal_()
    coef = torch.stack([0.5 + torch.rand(C1, generator=gen), 0.3 * torch.randn(C1, generator=gen),
                        0.2 * torch.randn(C1, generator=gen), 0.5 + torch.rand(C1, generator=gen)]).cuda().contiguous()
    Creal = C1 - 3                          # (the last three channels are padding: dx = 0 there)
    gamma = (0.5 + torch.rand(Creal, generator=gen)).cuda()
    st = rt.stream
    old = View.alloc(rt, N, H, W, C1)
    old.t.normal_()
    count = float(N * H * W)

    # reference: stored gradient + reduction, then the fused apply -- on the general kernel, which the two launches run on (the thin
    # kernel that normally serves the stored form adds the products in another order)
    g_ref = View.alloc(rt, N, H, W, C1)
    sums_ref = rt.zeros((16, 2, C1), torch.float64)
    nv.call('segnb_tune', b'fprop_thin', 0)
    try:
        op.dgrad(dyv, g_ref, bn_reduce=(y1, coef, sums_ref, act, 0.01))
        torch.cuda.synchronize()
    finally:
        nv.call('segnb_tune', b'fprop_thin', 1)
    dx_ref = View.alloc(rt, N, H, W, C1)
    dx_ref.t.copy_(old.t)
    bc_ref, dg_ref, db_ref = rt.zeros((3, C1), torch.float32), torch.ones(Creal, device='cuda'), torch.ones(Creal, device='cuda')
    nv.call('segnb_bn_bwd_apply_fused_direct_acc' if accumulate else 'segnb_bn_bwd_apply_fused_direct', rt.code, y1.ptr, y1.ld,
            N, H, W, Creal, C1, nv.ptr(coef), nv.ptr(sums_ref), nv.ptr(gamma), nv.ptr(bc_ref), nv.ptr(dg_ref), nv.ptr(db_ref), 1, None,
            act, 0.01, g_ref.ptr, g_ref.ld, dx_ref.ptr, dx_ref.ld, st)

    # the two launches
    assert op.dgrad_bnapply_ok(dyv, N, H, W, y1.ld)
    sums = rt.zeros((16, 2, C1), torch.float64)
    op.dgrad_bnsums(dyv, H, W, (y1, coef, sums, act, 0.01))
    dx = View.alloc(rt, N, H, W, C1)
    dx.t.copy_(old.t)
    bc, dg, db = rt.zeros((3, C1), torch.float32), torch.ones(Creal, device='cuda'), torch.ones(Creal, device='cuda')
    ep = nv.BnApplyEpilogue(y1.ptr, y1.ld, nv.ptr(coef), nv.ptr(sums), nv.ptr(gamma), Creal, count, nv.ptr(bc), nv.ptr(dg), nv.ptr(db),
                            act, 0.01, dx.ptr, dx.ld, accumulate)
    op.dgrad_bnapply(dyv, H, W, ep)
    torch.cuda.synchronize()
    a, b = sums.sum(0).cpu().numpy(), sums_ref.sum(0).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9 * float(np.abs(b).max()))       # (the same code, another block order)
    assert torch.equal(sums.sum(0), sums.sum(0)) and float(sums.abs().sum()) > 0           # (read, not cleared)
    np.testing.assert_allclose(bc.cpu().numpy(), bc_ref.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dg.cpu().numpy(), dg_ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), db_ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    x, r = dx.t.float(), dx_ref.t.float()
    diff = (x - r).abs()
    nbad = int((diff > 0).sum())
    assert nbad <= max(4, x.numel() // 2000), (nbad, x.numel())
    assert float(diff.max()) <= 2.0 ** -6 * float(r.abs().max()) + 1e-6
    assert float(x.view(N, H, W, C1)[..., Creal:].abs().max()) == (float(old.t.float().view(N, H, W, C1)[..., Creal:].abs().max()) if accumulate else 0.0)
    if N * H * W <= 20000:
        def run_emu():
            r_ = Runtime('cpu', 'bf16')
            ope = ConvOp(r_, w2.cpu(), None, [(C1, C1)], 1, 1, False, True)
            ope.pack(H, W)
            dye, ye, dxe = View.alloc(r_, N, H, W, ope.Cop), View.alloc(r_, N, H, W, C1), View.alloc(r_, N, H, W, C1)
            dye.t.copy_(dyv.t.cpu()); ye.t.copy_(y1.t.cpu()); dxe.t.copy_(old.t.cpu())
            se, ce, ge = r_.zeros((16, 2, C1), torch.float64), coef.cpu(), gamma.cpu()
            ope.dgrad_bnsums(dye, H, W, (ye, ce, se, act, 0.01))
            bce, dge, dbe = r_.zeros((3, C1), torch.float32), torch.ones(Creal), torch.ones(Creal)
            epe = nv.BnApplyEpilogue(ye.ptr, ye.ld, nv.ptr(ce), nv.ptr(se), nv.ptr(ge), Creal, count, nv.ptr(bce), nv.ptr(dge),
                                     nv.ptr(dbe), act, 0.01, dxe.ptr, dxe.ld, accumulate)
            ope.dgrad_bnapply(dye, H, W, epe)
            return dxe.t.float(), dge, dbe
        with on_emulator():
            dx_e, dg_e, db_e = run_emu()
        check('dx vs emulator', dx.t, dx_e, 'bf16')
        np.testing.assert_allclose(dg.cpu().numpy(), dg_e.numpy(), rtol=2e-3, atol=2e-3 * float(dg_e.abs().max()))
        np.testing.assert_allclose(db.cpu().numpy(), db_e.numpy(), rtol=2e-3, atol=2e-3 * float(db_e.abs().max()))


@pytest.mark.parametrize('shape', [(2, 24, 40, 112, 16, 1), (3, 9, 11, 48, 16, 2), (8, 8, 8, 1072, 16, 1), (8, 16, 16, 656, 16, 1),
                                   (2, 33, 17, 200, 8, 2), (8, 128, 128, 272, 16, 1), (8, 256, 256, 112, 16, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_thin_kernel_vs_general(shape):
    """conv_thin_kernel (fprop_thin.hip: <= 16 -> >= 48 channels, a dense layer's data gradient, tiramisu.py:9-20) against the
    ORACLE -- F.conv2d's autograd on the CPU for the gradient, torch's activation backward + float64 sums for the fused
    BatchNorm-backward reduction -- and then against its siblings: the general gather kernel (bf16 rounding of another summation
    order), the reduction pass run on ITS output; plain and fused launches store the same bits; channel splits (few pixels)."""
    N, H, W, C1, C2, act = shape
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 5 + C1)
    w2 = (torch.randn(C2, C1, 3, 3, generator=gen) * (2.0 / (C1 * 9)) ** 0.5).cuda()
    op = ConvOp(rt, w2, None, [(C1, C1)], 1, 1, False, True)
    op.pack(H, W)
    dyv = View.alloc(rt, N, H, W, op.Cop)
    dyv.t.normal_()
    y1 = View.alloc(rt, N, H, W, C1)
    y1.t.normal_()
    coef = torch.stack([0.5 + torch.rand(C1, generator=gen), 0.3 * torch.randn(C1, generator=gen),
                        0.2 * torch.randn(C1, generator=gen), 0.5 + torch.rand(C1, generator=gen)]).cuda().contiguous()
    st = rt.stream
    g_gen = View.alloc(rt, N, H, W, C1)
    nv.call('segnb_tune', b'fprop_thin', 0)
    try:
        op.dgrad(dyv, g_gen)
        torch.cuda.synchronize()
    finally:
        nv.call('segnb_tune', b'fprop_thin', 1)
    g_plain, g_f = View.alloc(rt, N, H, W, C1), View.alloc(rt, N, H, W, C1)
    g_plain.t.fill_(3.0); g_f.t.fill_(5.0)
    op.dgrad(dyv, g_plain)
    sums_f, sums_f2 = rt.zeros((16, 2, C1), torch.float64), rt.zeros((16, 2, C1), torch.float64)
    op.dgrad(dyv, g_f, bn_reduce=(y1, coef, sums_f, act, 0.01))
    op.dgrad(dyv, g_f, bn_reduce=(y1, coef, sums_f2, act, 0.01))
    sums_ref = rt.zeros((16, 2, C1), torch.float64)
    nv.call('segnb_bn_act_bwd_reduce', rt.code, y1.ptr, y1.ld, N, H, W, C1, nv.ptr(coef), act, 0.01, None, g_plain.ptr, g_plain.ld,
            None, 0, None, 0, None, 0, nv.ptr(sums_ref), None, 0, st)
    torch.cuda.synchronize()
    # the oracle first: the data gradient is F.conv2d's own autograd on the CPU (fp32 on the same bf16 operands) ...
    xr = torch.zeros(N, C1, H, W, requires_grad=True)
    wr = w2.detach().cpu().bfloat16().float()
    F.conv2d(xr, wr, None, padding=1).backward(dyv.dense()[..., :C2].float().cpu().permute(0, 3, 1, 2))
    check('thin vs F.conv2d backward', g_plain.dense().float().cpu().permute(0, 3, 1, 2), xr.grad, 'bf16')
    # ... and the fused reduction's sums are torch's, taken over the gradient the launch STORED (the epilogue's own arithmetic:
    # dz through the activation of the producing BatchNorm, edz / eydz of lib/modules/abn/functions.py:107-112)
    _, s1_t, s2_t = torch_bn_act_bwd(y1.dense(), g_plain.dense(), coef, act, 0.01, 'bf16')
    a = sums_f.sum(0).cpu().numpy()
    for got, ref_t in ((a[0], s1_t.numpy()), (a[1], s2_t.numpy())):
        assert np.abs(got - ref_t).max() <= 2e-5 * float(np.abs(ref_t).max()) + 1e-6 * (N * H * W) ** 0.5, np.abs(got - ref_t).max()
    # then the siblings: the general gather kernel, the plain launch, the stand-alone reduction pass, a second run
    check('thin vs general', g_plain.t, g_gen.t, 'bf16')
    assert torch.equal(g_plain.t, g_f.t)
    b = sums_ref.sum(0).cpu().numpy()
    assert np.abs(a - b).max() <= 2e-5 * float(np.abs(b).max()) + 1e-6 * (N * H * W) ** 0.5, np.abs(a - b).max()
    np.testing.assert_allclose(sums_f2.sum(0).cpu().numpy(), a, rtol=1e-12, atol=1e-9)       # fixed order inside a block


@pytest.mark.parametrize('dtype', DTYPES)
def test_maxpool_bwd_two_sources(dtype):
    """segnb_maxpool_bwd_add(g1, g2) against F.max_pool2d's backward of round(g1 + g2) on the CPU (the oracle), and == segnb_add(g1, g2)
    followed by segnb_maxpool_bwd, bit for bit (the ResNet stem's MaxPool2d(3, 2, 1) whose output has two consumers, linknet.py:41-62)."""
    rt = Runtime('cuda', dtype)
    N, H, W, C = 2, 37, 41, 24
    x = View.alloc(rt, N, H, W, C)
    x.t.normal_()
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    o, g1, g2, gs = (View.alloc(rt, N, Ho, Wo, C) for _ in range(4))
    g1.t.normal_(); g2.t.normal_()
    idx = torch.zeros(N, Ho, Wo, C, dtype=torch.uint8, device='cuda')
    st = rt.stream
    nv.call('segnb_maxpool_fwd', rt.code, x.ptr, x.ld, N, H, W, C, 3, 2, 1, o.ptr, o.ld, nv.ptr(idx), st)
    nv.call('segnb_add', rt.code, g1.ptr, g1.ld, g2.ptr, g2.ld, gs.ptr, gs.ld, N, Ho, Wo, C, st)
    dx_ref, dx = View.alloc(rt, N, H, W, C), View.alloc(rt, N, H, W, C)
    nv.call('segnb_maxpool_bwd', rt.code, x.ptr, x.ld, gs.ptr, gs.ld, N, H, W, C, 3, 2, 1, dx_ref.ptr, dx_ref.ld, nv.ptr(idx), st)
    nv.call('segnb_maxpool_bwd_add', rt.code, x.ptr, x.ld, g1.ptr, g1.ld, g2.ptr, g2.ld, N, H, W, C, 3, 2, 1, dx.ptr, dx.ld,
            nv.ptr(idx), st)
    torch.cuda.synchronize()
    # oracle: F.max_pool2d's backward of round(g1 + g2) on the CPU
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    xr = x.dense().float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    gsum = (g1.dense().float().cpu() + g2.dense().float().cpu()).to(tdt).float()
    F.max_pool2d(xr, 3, 2, 1).backward(gsum.permute(0, 3, 1, 2))
    check('vs torch', dx.dense().float().cpu().permute(0, 3, 1, 2), xr.grad, dtype)
    assert torch.equal(dx.t, dx_ref.t)        # sibling: the two launches it replaces, bit for bit


ACT_EP_CASES = [
    # name,                 N, H,  W,  segs,               Co, k, s, p, transposed
    ('ws 64->64 relu',      2, 24, 40, [(64, 64)],         64, 3, 1, 1, False),
    ('ws 128->72 leaky',    1, 33, 17, [(128, 128)],       72, 3, 1, 1, False),
    ('c8 3->32',            2, 32, 48, [(3, 8)],           32, 3, 1, 1, False),
    ('c8 3->64 halves',     2, 32, 48, [(3, 8)],           64, 3, 1, 1, False),     # unet16.py:73 (VGG's first convolution)
    ('c8 3->40 halves leaky', 1, 19, 27, [(3, 8)],         40, 3, 1, 1, False),
    ('rw 32->64 relu',      2, 24, 40, [(32, 32)],         64, 3, 1, 1, False),
    ('rw cat 96->32 leaky', 1, 30, 36, [(64, 64), (32, 32)], 32, 3, 1, 1, False),
    ('general 1x1 40->24',  2, 19, 23, [(40, 40)],         24, 1, 1, 0, False),
    ('general 3x3 s2',      2, 21, 30, [(16, 16)],         48, 3, 2, 1, False),
    ('general 2x2 p1',      1, 15, 15, [(32, 32)],         8,  2, 1, 1, False),
    # the parity phases of a transposed convolution, each with the epilogue on its own outputs (linknet.py:58 finaldeconv1)
    ('phases T3x3 s2 leaky', 2, 12, 14, [(64, 64)],        32, 3, 2, 0, True),
    # ConvTranspose2d(4, 2, 1) -> ReLU (unet16.py:38-40): the four phases as one launch with the activation in its staging
    ('upconv T4x4 s2 relu', 2, 16, 24, [(128, 128)],       64, 4, 2, 1, True),
    ('upconv T4x4 s2 leaky', 1, 13, 17, [(192, 192)],      40, 4, 2, 1, True),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('with_bn', [False, True], ids=['act', 'evalbn+act'])
@pytest.mark.parametrize('case', ACT_EP_CASES, ids=[c[0] for c in ACT_EP_CASES])
def test_conv_fprop_act_epilogue(case, with_bn, dtype):
    """segnb_conv_fprop_act: activation (and eval-mode BatchNorm) in the convolution's epilogue == conv -> [BatchNorm2d in
    eval mode] -> ReLU / LeakyReLU as torch computes them (unet16.py:12-21, linknet.py:57-62; validate() of
    torch_train.py:248-265), and == the emulator.  The output is written into a channel slice of a wider buffer."""
    name, N, H, W, segs, Co, k, s, p, transposed = case
    Ci = sum(r for r, _ in segs)
    gen = torch.Generator().manual_seed(len(name) * 7 + Co)
    rt = Runtime('cuda', dtype)
    q = (lambda t: t.bfloat16().float()) if dtype == 'bf16' else (lambda t: t)
    w = q(torch.randn(Co, Ci, k, k, generator=gen) * (2.0 / (Ci * k * k)) ** 0.5)
    b = torch.randn(Co, generator=gen) * 0.1
    x = q(torch.randn(N, Ci, H, W, generator=gen))
    act, slope = (nv.ACT_LEAKY, 0.01) if 'leaky' in name or k == 2 else (nv.ACT_RELU, 0.0)
    bn = torch.nn.BatchNorm2d(Co).eval()
    with torch.no_grad():
        bn.weight.copy_(0.5 + torch.rand(Co, generator=gen)); bn.bias.copy_(0.2 * torch.randn(Co, generator=gen))
        bn.running_mean.copy_(0.1 * torch.randn(Co, generator=gen)); bn.running_var.copy_(0.5 + torch.rand(Co, generator=gen))
    if transposed:
        w = q(torch.randn(Ci, Co, k, k, generator=gen) * (2.0 / (Ci * k * k)) ** 0.5)
    with torch.no_grad():
        ref = F.conv_transpose2d(x, w, b, stride=s, padding=p) if transposed else F.conv2d(x, w, b, stride=s, padding=p)
        if with_bn:
            ref = bn(ref)
        ref = F.leaky_relu(ref, slope) if act == nv.ACT_LEAKY else torch.relu(ref)

    def run(device):
        r = Runtime(device, dtype)
        op = ConvOp(r, w.to(device), b.to(device), segs, s, p, transposed, need_dgrad=False)
        op.pack(H, W)
        Ho, Wo = op.out_hw(H, W)
        xv = View.alloc(r, N, H, W, op.Cip)
        off, roff = 0, 0
        for real, padded in segs:
            xv.dense()[..., off:off + real] = x[:, roff:roff + real].permute(0, 2, 3, 1).to(device, r.tdtype)
            off += padded
            roff += real
        buf = r.zeros((N, Ho, Wo, op.Cop + 16))
        yv = View(buf, N, Ho, Wo, op.Cop, op.Cop + 16, 8)
        coef = None
        if with_bn:
            coef = r.zeros((4, op.Cop), torch.float32)
            stats = r.zeros((16, 2, op.Cop), torch.float64)
            dv = lambda t: t.detach().to(device)
            rm, rv, gam, bet = dv(bn.running_mean), dv(bn.running_var), dv(bn.weight), dv(bn.bias)     # (kept alive)
            nv.call('segnb_bn_finalize', nv.ptr(stats), Co, op.Cop, float(N * Ho * Wo), nv.ptr(gam), nv.ptr(bet), 1e-5,
                    0.1, nv.ptr(rm), nv.ptr(rv), None, 0, nv.ptr(coef), r.stream)
            if device != 'cpu':
                torch.cuda.synchronize()
        if 'upconv' in name:
            if with_bn:
                assert not op.act_epilogue_ok(H, W, N, yv.ld, coef)     # (no folded BatchNorm on that kernel)
                return None, None
            assert op.act_epilogue_ok(H, W, N, yv.ld, coef)
        else:
            assert op.act_epilogue_ok(H, W)
        op.fprop(xv, yv, None, epilogue=(coef, act, slope))
        if device != 'cpu':
            torch.cuda.synchronize()
        return yv.dense().float().cpu(), buf.float().cpu()

    if 'upconv' in name and dtype != 'bf16':
        pytest.skip('segnb_upconv_fprop is bf16 only')
    y_g, buf_g = run('cuda')
    if y_g is None:
        return
    with on_emulator():
        y_e, _ = run('cpu')
    check(name + ' vs emulator', y_g, y_e, dtype)
    check(name + ' vs torch', y_g[..., :Co].permute(0, 3, 1, 2), ref, dtype)
    assert float(buf_g[..., :8].abs().max()) == 0.0 and float(buf_g[..., 8 + y_g.shape[-1]:].abs().max()) == 0.0
    if y_g.shape[-1] > Co:
        pad = y_g[..., Co:]
        assert float(pad.abs().max()) == 0.0, 'padding channels must stay zero'


# ------------------------------------------------------------------------------------------------------
# resident-weights pipeline of the thin layers (fprop_rw.hip): Ci in {32, 64, 96}, Co <= 96
# ------------------------------------------------------------------------------------------------------
RW_CASES = [
    # name,             N, H,  W,  segs,                    Co
    ('rw 32->32',       2, 21, 37, [(32, 32)],              32),     # ragged tiles
    ('rw 96->32',       3, 33, 17, [(64, 64), (32, 32)],    32),     # 3 chunks; data gradient 32 -> 96 (12 chunks / row)
    ('rw 32->64',       1, 30, 30, [(32, 32)],              64),
    ('rw 64->24',       2, 16, 56, [(64, 64)],              24),     # 2 chunks, padded output channels
    ('rw 32->40',       5, 40, 24, [(32, 32)],              40),     # more tiles than blocks' first round
    ('rw cat 62->32',   1, 19, 23, [(30, 32), (32, 32)],    32),     # padded concat segment
]


@pytest.mark.parametrize('case', RW_CASES, ids=[c[0] for c in RW_CASES])
def test_conv_fprop_rw(case):
    name, N, H, W, segs, Co = case
    full = (name, N, H, W, segs, Co, 3, 1, 1, False)
    Ci = sum(r for r, _ in segs)
    gen = torch.Generator().manual_seed(11)
    w = (torch.randn((Co, Ci, 3, 3), generator=gen) * (2.0 / (Ci * 9)) ** 0.5).bfloat16().float()
    b = torch.randn(Co, generator=gen) * 0.1
    x = torch.randn(N, Ci, H, W, generator=gen).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=gen).bfloat16().float()
    y_g, st_g, dx_g, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
    y_g2, _, dx_g2, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
    nv.call('segnb_tune', b'fprop_rw', 0)
    try:
        y_o, st_o, dx_o, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
    finally:
        nv.call('segnb_tune', b'fprop_rw', 1)
    with on_emulator():
        y_e, st_e, dx_e, _, _, _ = _run_conv('cpu', 'bf16', full, w, b, x, dy)
    check(name + ' y', y_g, y_e, 'bf16')
    check(name + ' dx', dx_g, dx_e, 'bf16')
    check(name + ' y vs other kernels', y_g, y_o, 'bf16')
    check(name + ' dx vs other kernels', dx_g, dx_o, 'bf16')
    np.testing.assert_allclose(st_g.numpy(), st_e.numpy(), rtol=2e-3, atol=2e-2 * float(st_e.abs().max()))
    assert float(y_g[..., Co:].abs().max()) == 0.0 if y_g.shape[-1] > Co else True
    assert torch.equal(y_g, y_g2) and torch.equal(dx_g, dx_g2)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, b, padding=1)
    yr.backward(dy)
    check(name + ' y vs torch', y_g[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
    parts, off = [], 0
    for real, padded in segs:
        parts.append(dx_g[..., off:off + real])
        off += padded
    check(name + ' dx vs torch', torch.cat(parts, -1).permute(0, 3, 1, 2), xr.grad, 'bf16')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 5, 7, 16), (1, 1, 9, 8), (3, 16, 16, 40), (4, 128, 128, 64)], ids=lambda s: 'x'.join(map(str, s)))
def test_upsample_bilinear2x_vs_torch(shape, dtype):
    """segnb_upsample_bilinear2x_fwd / _bwd == nn.Upsample(scale_factor=2, mode='bilinear') (lib/models/unet16.py:43) and its
    autograd backward on the CPU (fp32 math on the same rounded operands); one-row / one-column images hit the clamped taps."""
    N, H, W, C = shape
    rt = Runtime('cuda', dtype)
    gen = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(N, C, H, W, generator=gen)
    go = torch.randn(N, C, 2 * H, 2 * W, generator=gen)
    if dtype == 'bf16':
        x, go = x.bfloat16().float(), go.bfloat16().float()
    xv = View.alloc(rt, N, H, W, C)
    xv.dense().copy_(x.permute(0, 2, 3, 1).to('cuda', rt.tdtype))
    ov = View.alloc(rt, N, 2 * H, 2 * W, C)
    gv = View.alloc(rt, N, 2 * H, 2 * W, C)
    gv.dense().copy_(go.permute(0, 2, 3, 1).to('cuda', rt.tdtype))
    dxv = View.alloc(rt, N, H, W, C)
    nv.call('segnb_upsample_bilinear2x_fwd', rt.code, xv.ptr, xv.ld, N, H, W, C, ov.ptr, ov.ld, rt.stream)
    nv.call('segnb_upsample_bilinear2x_bwd', rt.code, gv.ptr, gv.ld, N, H, W, C, dxv.ptr, dxv.ld, rt.stream)
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=False)
    yr.backward(go)
    check('bilinear fwd', ov.dense().float().cpu().permute(0, 3, 1, 2), yr.detach(), dtype)
    check('bilinear bwd', dxv.dense().float().cpu().permute(0, 3, 1, 2), xr.grad, dtype)


def test_fork_carried_by_the_apply_pass_orders_the_side_stream():
    """segnb_stream_fork_arm / segnb_stream_fork_commit: the side stream waits for the BatchNorm-backward apply pass through an
    event that rides on that kernel's own dispatch (no marker packet on the main queue) -- the weight gradient launched behind
    every apply pass (zf_unet.py's backward through segnb.engine.Stage.backward) must see the finished dy.  A reader on the side
    stream sums dy right behind the commit, 40 times with a different gradient each time; a reader that started early would see
    values of the previous round.  Also: arm with NO carrying launch in between (commit falls back to an ordinary fork)."""
    rt = Runtime('cuda', 'bf16')
    N, H, W, C = 32, 112, 112, 64
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    y = View.alloc(rt, N, H, W, C)
    g = View.alloc(rt, N, H, W, C)
    dy = View.alloc(rt, N, H, W, C)
    coef = rt.zeros((4, C), torch.float32)
    coef[0] = 1.0
    coef[3] = 1.0
    bcoef = rt.zeros((3, C), torch.float32)
    bcoef[0] = 1.0                                   # dy = 1 * (g - 0 - yhat * 0) = g
    n = N * H * W * C
    got = []
    for it in range(1, 41):
        carried = it % 4 != 0
        g.t.fill_(float(it % 7 + 1))
        nv.call('segnb_stream_fork_arm', rt.stream)
        if carried:
            nv.call('segnb_bn_bwd_apply_direct', rt.code, y.ptr, y.ld, N, H, W, C, nv.ptr(coef), nv.ptr(bcoef), nv.ACT_NONE, 0.0,
                    g.ptr, g.ld, dy.ptr, dy.ld, None, C, rt.stream)
        else:
            nv.call('segnb_add', rt.code, None, 0, g.ptr, g.ld, dy.ptr, dy.ld, N, H, W, C, rt.stream)      # (cannot carry an event)
        nv.call('segnb_stream_fork_commit', rt.stream, side.cuda_stream)
        with torch.cuda.stream(side):
            got.append((it, dy.t.sum(dtype=torch.float64)))
        main.wait_stream(side)                       # the next round's fill must not overtake the reader
    torch.cuda.synchronize()
    for it, v in got:
        assert float(v) == float(n * (it % 7 + 1)), (it, float(v))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 9, 11, 32, nv.ACT_RELU), (3, 8, 8, 72, nv.ACT_LEAKY)], ids=['relu', 'leaky'])
def test_bn_bwd_apply_direct_equals_two_pass(shape, dtype):
    """sums-only reduce (dz == NULL) + segnb_bn_bwd_apply_direct == reduce (dz stored) + segnb_bn_bwd_apply, bit for
    bit on the GPU and on the emulator, and GPU == emulator within the elementwise tolerance."""
    N, H, W, C, act = shape
    Cp = cp.pad8(C)
    gen = torch.Generator().manual_seed(5)
    y0 = torch.randn(N, H, W, C, generator=gen)
    g0 = torch.randn(N, H, W, C, generator=gen)
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen) * 0.2

    def run(device, direct):
        rt = Runtime(device, dtype)
        dev = rt.device
        yv = View.alloc(rt, N, H, W, Cp)
        yv.dense()[..., :C] = y0.to(dev, rt.tdtype)
        gv = View.alloc(rt, N, H, W, Cp)
        gv.dense()[..., :C] = g0.to(dev, rt.tdtype)
        stats = torch.zeros(16, 2, Cp, dtype=torch.float64, device=dev)
        yy = yv.dense().double()
        stats[0, 0] = yy.sum((0, 1, 2))
        stats[0, 1] = (yy * yy).sum((0, 1, 2))
        coef = rt.zeros((4, Cp), torch.float32)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        g_, b_ = gamma.to(dev), beta.to(dev)
        nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * H * W), nv.ptr(g_), nv.ptr(b_), 1e-5, 0.1,
                nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), 1, nv.ptr(coef), rt.stream)
        dz = View.alloc(rt, N, H, W, Cp)
        sums = rt.zeros((16, 2, Cp), torch.float64)
        nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), act, 0.01, None,
                gv.ptr, gv.ld, None, 0, None, 0, None if direct else dz.ptr, dz.ld, nv.ptr(sums), None, 0, rt.stream)
        sums_copy = sums.sum(0)
        bcoef = rt.zeros((3, Cp), torch.float32)
        dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        nv.call('segnb_bn_bwd_finalize', nv.ptr(sums), C, Cp, float(N * H * W), nv.ptr(g_), nv.ptr(coef), nv.ptr(bcoef),
                nv.ptr(dgam), nv.ptr(dbet), 0, rt.stream)
        dyv = View.alloc(rt, N, H, W, Cp)
        if direct:
            nv.call('segnb_bn_bwd_apply_direct', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), act,
                    0.01, gv.ptr, gv.ld, dyv.ptr, dyv.ld, None, C, rt.stream)
        else:
            nv.call('segnb_bn_bwd_apply', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), nv.ptr(bcoef), dz.ptr, dz.ld,
                    dyv.ptr, dyv.ld, None, C, rt.stream)
        if device != 'cpu':
            torch.cuda.synchronize()
        return dyv.dense().float().cpu(), sums_copy.cpu(), dgam.cpu()

    dy_a, s_a, dg_a = run('cuda', False)
    dy_b, s_b, dg_b = run('cuda', True)
    assert torch.equal(dy_a, dy_b) and torch.equal(s_a, s_b) and torch.equal(dg_a, dg_b)
    with on_emulator():
        dy_e, s_e, _ = run('cpu', False)
        dy_f, s_f, _ = run('cpu', True)
    assert torch.equal(dy_e, dy_f) and torch.equal(s_e, s_f)
    check('apply_direct dy', dy_b, dy_f, dtype)


@pytest.mark.parametrize('case', [('tall 7x7 a', 5, 7, 7, [(128, 128)], 72), ('tall 7x7 b', 9, 7, 7, [(64, 64)], 200),
                                  ('tall 7x7 cat', 3, 7, 7, [(64, 64), (64, 64)], 64)], ids=lambda c: c[0])
def test_conv_fprop_dma_tall_7x7(case):
    """tall-image tiles of the 7x7 level (fprop_dma.hip, WsCfg<..., TALL>): images chained with one shared zero row;
    ragged last tile, several channel tiles, two chunks."""
    name, N, H, W, segs, Co = case
    full = (name, N, H, W, segs, Co, 3, 1, 1, False)
    Ci = sum(r for r, _ in segs)
    gen = torch.Generator().manual_seed(13)
    w = (torch.randn((Co, Ci, 3, 3), generator=gen) * (2.0 / (Ci * 9)) ** 0.5).bfloat16().float()
    b = torch.randn(Co, generator=gen) * 0.1
    x = torch.randn(N, Ci, H, W, generator=gen).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=gen).bfloat16().float()
    y_g, st_g, dx_g, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
    y_g2, _, _, _, _, _ = _run_conv('cuda', 'bf16', full, w, b, x, dy)
    with on_emulator():
        y_e, st_e, dx_e, _, _, _ = _run_conv('cpu', 'bf16', full, w, b, x, dy)
    check(name + ' y', y_g, y_e, 'bf16')
    check(name + ' dx', dx_g, dx_e, 'bf16')
    np.testing.assert_allclose(st_g.numpy(), st_e.numpy(), rtol=2e-3, atol=2e-2 * float(st_e.abs().max()))
    assert torch.equal(y_g, y_g2)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, b, padding=1)
    yr.backward(dy)
    check(name + ' y vs torch', y_g[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')


@pytest.mark.parametrize('shape', [(32, 512, 1024), (32, 1024, 1024), (32, 1024, 512), (5, 256, 72), (9, 512, 200)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_fprop_split_k_7x7(shape):
    """Split K of conv_fprop_ws_kernel at the 7 x 7 level (lib/models/zf_unet.py:50, 512 / 1024 channels): KS blocks share one
    tile, publish fp32 slabs, the block whose ticket is last adds them in slice order.  Against F.conv2d on the CPU (the oracle);
    KS = 2 / 4 against the unsplit launch (same bf16 outputs but for rounding-boundary flips, same statistics); bitwise equal
    from launch to launch -- also with a second stream keeping half of the chip busy and the caches warm or flushed, the
    conditions under which a stale slab would show."""
    N, Ci, Co = shape
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(Ci + Co + N)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) * (2.0 / (Ci * 9)) ** 0.5).bfloat16().float()
    b = torch.randn(Co, generator=gen) * 0.1
    x = torch.randn(N, Ci, 7, 7, generator=gen).bfloat16().float()
    op = ConvOp(rt, w.cuda(), b.cuda(), [(Ci, Ci)], 1, 1, False, need_dgrad=False)
    op.pack(7, 7)
    xv = View.alloc(rt, N, 7, 7, op.Cip)
    xv.dense()[..., :Ci] = x.permute(0, 2, 3, 1).to('cuda', torch.bfloat16)
    yr = F.conv2d(x, w, b, padding=1)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device='cuda')
    res = {}
    try:
        for ks in (0, 1, 2, 4):
            nv.call('segnb_tune', b'fprop_ksplit', ks)
            outs = []
            for rep in range(6):
                yv = View.alloc(rt, N, 7, 7, op.Cop)
                yv.t.fill_(7.0)
                stats = rt.zeros((16, 2, op.Cop), torch.float64)
                if rep % 3 == 1:      # cold caches
                    flush = torch.empty(160 << 20, dtype=torch.float32, device='cuda').fill_(1.0)
                    del flush
                if rep >= 3:          # uneven load: another queue holds CUs while the slices arrive
                    with torch.cuda.stream(side):
                        for _ in range(4):
                            junk @ junk
                op.fprop(xv, yv, stats)
                torch.cuda.synchronize()
                outs.append((yv.dense().clone(), stats.sum(0).cpu()))
            for k in range(1, 6):
                assert torch.equal(outs[0][0], outs[k][0]), 'ks=%d: launch %d differs from launch 0' % (ks, k)
                np.testing.assert_allclose(outs[k][1].numpy(), outs[0][1].numpy(), rtol=1e-12)
            res[ks] = outs[0]
    finally:
        nv.call('segnb_tune', b'fprop_ksplit', 1)
    name = 'split-K 7x7 ' + 'x'.join(map(str, shape))
    for ks in (0, 1, 2, 4):
        y_g = res[ks][0].float().cpu()
        check('%s ks=%d y vs torch' % (name, ks), y_g[..., :Co].permute(0, 3, 1, 2), yr, 'bf16')
        assert op.Cop == Co or float(y_g[..., Co:].abs().max()) == 0.0
        ys = yr.bfloat16().double()
        st = res[ks][1]
        np.testing.assert_allclose(st[0, :Co].numpy(), ys.sum((0, 2, 3)).numpy(), rtol=1e-3,
                                   atol=2e-2 * float(ys.abs().max()) * (N * 49) ** 0.5)
        np.testing.assert_allclose(st[1, :Co].numpy(), (ys * ys).sum((0, 2, 3)).numpy(), rtol=2e-3)
    # the split sum is another fp32 evaluation order: outputs equal the unsplit launch's but for single-ulp flips (near zero:
    # the absolute fp32 difference of a cancelling sum)
    for ks in (2, 4):
        got, ref = res[ks][0].float().cpu(), res[0][0].float().cpu()
        diff = (got - ref).abs()
        assert bool((diff <= ref.abs() * 2.0 ** -7 + 2e-5).all()), '%s ks=%d: more than one bf16 ulp from the unsplit launch' % (name, ks)
        assert int((diff > 0).sum()) <= 2e-2 * got.numel()


# ------------------------------------------------------------------------------------------------------
# operands recomputed on load (segnb_operand_tf): rolling-window kernels of fprop_roll.hip / wgrad_roll.hip
# ------------------------------------------------------------------------------------------------------
def _ulp_close(name, got, ref, frac=2e-3):
    """bf16 tensors produced by two evaluation orders of the same fp32 expression: equal but for single-ulp flips of a
    small fraction of the elements (a rounding boundary between the two fp32 values)"""
    got, ref = got.float().cpu(), ref.float().cpu()
    diff = (got - ref).abs()
    ulp = ref.abs().clamp_min(1e-30) * 2.0 ** -7
    assert bool((diff <= ulp + 1e-30).all()), '%s: more than one bf16 ulp apart (max %g)' % (name, float(diff.max()))
    nbad = int((diff > 0).sum())
    assert nbad <= frac * got.numel() + 4, '%s: %d of %d elements differ' % (name, nbad, got.numel())


@pytest.mark.parametrize('shape', [(2, 40, 56, 32, 1), (3, 33, 47, 24, 2), (32, 224, 224, 32, 1)], ids=lambda s: 'x'.join(map(str, s)))
def test_conv_with_operands_recomputed_on_load(shape):
    """conv3x3(act(BatchNorm(y0))) forward, its data gradient and its weight gradient with the activated input and dy never in
    memory (segnb_conv_fprop_tf / segnb_conv_wgrad_tf) against the launches they replace: segnb_bn_act_fwd + segnb_conv_fprop,
    segnb_bn_bwd_apply_direct + segnb_conv_fprop_bnreduce, segnb_conv_wgrad -- and against the emulator's restatement."""
    N, H, W, C2, act = shape
    C1 = 32
    rt = Runtime('cuda', 'bf16')
    gen = torch.Generator().manual_seed(H * 3 + C2)
    w2 = (torch.randn(C2, C1, 3, 3, generator=gen) * (2.0 / (C1 * 9)) ** 0.5).cuda()
    b2 = (torch.randn(C2, generator=gen) * 0.1).cuda()
    op = ConvOp(rt, w2, b2, [(C1, C1)], 1, 1, False, True)
    op.pack(H, W)
    C2p = op.Cop
    y0 = View.alloc(rt, N, H, W, C1)          # pre-BatchNorm output of the producing layer
    y0.t.normal_()
    y0b = View.alloc(rt, N, H, W, C1)         # pre-BatchNorm output of the layer before THAT one (BatchNorm-reduce epilogue)
    y0b.t.normal_()
    mk = lambda C: torch.stack([0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen),
                                0.2 * torch.randn(C, generator=gen), 0.5 + torch.rand(C, generator=gen)]).cuda().contiguous()
    coef0, coef2 = mk(C1), mk(C2p)
    coef2[:, C2:] = 0
    bcoef2 = torch.stack([coef2[0], 0.05 * torch.randn(C2p, generator=gen).cuda(), 0.05 * torch.randn(C2p, generator=gen).cuda()]).contiguous()
    bcoef2[:, C2:] = 0
    drop = (torch.rand(N, C1, generator=gen) > 0.3).float().mul(1.0 / 0.7).cuda().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    # ---- forward: reference = activation pass + convolution with statistics
    a0 = View.alloc(rt, N, H, W, C1)
    nv.call('segnb_bn_act_fwd', rt.code, y0.ptr, y0.ld, N, H, W, C1, nv.ptr(coef0), act, 0.01, nv.ptr(drop), a0.ptr, a0.ld,
            None, 0, None, 0, None, 0, st)
    y2r = View.alloc(rt, N, H, W, C2p)
    sr = rt.zeros((16, 2, C2p), torch.float64)
    op.fprop(a0, y2r, sr)
    assert op.fprop_tf_ok(y0, y2r)
    y2 = View.alloc(rt, N, H, W, C2p)
    s2 = rt.zeros((16, 2, C2p), torch.float64)
    op.fprop_tf(y0, ConvOp.tf_act(coef0, C1, act, 0.01, drop), y2, s2)
    torch.cuda.synchronize()
    # the activation is evaluated in a folded fp32 form (drop * scale and drop * (shift - mean * scale) per channel): its
    # bf16 result differs from the activation pass's by single-ulp flips of a small fraction of the elements
    check('y', y2.t, y2r.t, 'bf16')
    assert float((y2.t.float() - y2r.t.float()).abs().mean() / y2r.t.float().abs().mean()) < 1e-3
    sa, sb = s2.sum(0).cpu().numpy(), sr.sum(0).cpu().numpy()
    np.testing.assert_allclose(sa, sb, rtol=2e-3, atol=2e-3 * float(np.abs(sb).max()))
    # ---- backward: g = gradient of this layer's activation; reference = apply pass + data gradient (with reduce) + weight gradient
    g2 = View.alloc(rt, N, H, W, C2p)
    g2.t.normal_()
    g2.dense()[..., C2:] = 0
    dy = View.alloc(rt, N, H, W, C2p)
    nv.call('segnb_bn_bwd_apply_direct', rt.code, y2r.ptr, y2r.ld, N, H, W, C2p, nv.ptr(coef2), nv.ptr(bcoef2), act, 0.01,
            g2.ptr, g2.ld, dy.ptr, dy.ld, None, C2, st)
    dxr = View.alloc(rt, N, H, W, C1)
    sums_r = rt.zeros((16, 2, C1), torch.float64)
    if C2 == 32:
        op.dgrad(dy, dxr, bn_reduce=(y0b, coef0, sums_r, act, 0.01))
    else:
        op.dgrad(dy, dxr)
    gwr = torch.zeros_like(w2)
    op.wgrad(a0, dy, gwr)
    assert op.wgrad_tf_ok(y0, g2)
    has_dgrad = op.dgrad_tf_ok(g2, dxr)          # (the data gradient's K = this layer's output channels: 32 only)
    assert has_dgrad == (C2 == 32)
    tfd = ConvOp.tf_bnbwd(y2r, coef2, bcoef2, act, 0.01)
    dx = View.alloc(rt, N, H, W, C1)
    sums = rt.zeros((16, 2, C1), torch.float64)
    if has_dgrad:
        op.dgrad_tf(g2, tfd, dx, bn_reduce=(y0b, coef0, sums, act, 0.01))
    op.wgrad_tf(y0, ConvOp.tf_act(coef0, C1, act, 0.01, drop), g2, tfd)
    p = op.plan(H, W)
    gw = torch.zeros_like(w2)
    nv.call('segnb_unpack_wgrad', nv.ptr(p['dwp'][0]), nv.ptr(gw), op.Cop, op.Cip, 9, op.s_out, op.s_in, p['tapoff_fwd'][0],
            nv.ptr(op.out_map), nv.ptr(op.in_map), 1, st)
    torch.cuda.synchronize()
    # dy is evaluated in a folded fp32 form: the operand differs from the apply pass's by single bf16-ulp flips of a small
    # fraction of its elements, so dx / dW agree to that
    if has_dgrad:
        check('dx', dx.t, dxr.t, 'bf16')
        rel = float((dx.t.float() - dxr.t.float()).abs().mean() / dxr.t.float().abs().mean())
        assert rel < 2e-3, rel
        a_, b_ = sums.sum(0).cpu().numpy(), sums_r.sum(0).cpu().numpy()
        assert np.abs(a_ - b_).max() <= 2e-3 * np.abs(b_).max(), (np.abs(a_ - b_).max(), np.abs(b_).max())
    relw = float((gw - gwr).abs().max() / gwr.abs().max())
    assert relw < 3e-3, relw
    if N * H * W <= 20000:
        with on_emulator():
            cpu = lambda v: View(v.t.cpu(), v.N, v.H, v.W, v.Cp, v.ld, 0)
            rte = Runtime('cpu', 'bf16')
            ope = ConvOp(rte, w2.cpu(), b2.cpu(), [(C1, C1)], 1, 1, False, True)
            ope.pack(H, W)
            c0, c2, bc2, dr = coef0.cpu(), coef2.cpu(), bcoef2.cpu(), drop.cpu()
            y0e, y2re, g2e, y0be = cpu(y0), cpu(y2r), cpu(g2), cpu(y0b)
            y2e = View.alloc(rte, N, H, W, C2p)
            ope.fprop_tf(y0e, ConvOp.tf_act(c0, C1, act, 0.01, dr), y2e)
            check('y vs emulator', y2.t, y2e.t, 'bf16')
            dxe = View.alloc(rte, N, H, W, C1)
            se = torch.zeros((16, 2, C1), dtype=torch.float64)
            tfde = ConvOp.tf_bnbwd(y2re, c2, bc2, act, 0.01)
            if has_dgrad:
                ope.dgrad_tf(g2e, tfde, dxe, bn_reduce=(y0be, c0, se, act, 0.01))
            ope.wgrad_tf(y0e, ConvOp.tf_act(c0, C1, act, 0.01, dr), g2e, tfde)
            pe = ope.plan(H, W)
            gwe = torch.zeros_like(w2.cpu())
            nv.call('segnb_unpack_wgrad', nv.ptr(pe['dwp'][0]), nv.ptr(gwe), ope.Cop, ope.Cip, 9, ope.s_out, ope.s_in,
                    pe['tapoff_fwd'][0], nv.ptr(ope.out_map), nv.ptr(ope.in_map), 1, 0)
        if has_dgrad:
            check('dx vs emulator', dx.t, dxe.t, 'bf16')
        assert float((gw.cpu() - gwe).abs().max() / gwe.abs().max()) < 3e-3


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('srcs', ['d+drop', 'd+pool+drop', 'up+drop', 'd+pool', 'd+up'])
@pytest.mark.parametrize('shape', [(2, 10, 12, 24, nv.ACT_RELU), (3, 9, 7, 40, nv.ACT_LEAKY)], ids=['even relu', 'odd leaky'])
def test_bn_bwd_apply_from_sources_equals_stored_dz(shape, srcs, dtype):
    """segnb_bn_act_bwd_reduce(dz = NULL) + segnb_bn_bwd_apply_fused_src == reduce (dz stored) + segnb_bn_bwd_apply_fused, bit for
    bit, for every source combination of the ZF_UNET plan (direct + Dropout2d, skip + MaxPool2d routing + Dropout2d, Upsample
    sum): dz is never stored, the apply pass re-reads the sources (lib/models/zf_unet.py:25,31,41-42 backward)."""
    N, H, W, C, act = shape
    Cp = cp.pad8(C)
    gen = torch.Generator().manual_seed(11)
    y0 = torch.randn(N, H, W, C, generator=gen)
    gd0 = torch.randn(N, H, W, C, generator=gen)
    gp0 = torch.randn(N, H // 2, W // 2, C, generator=gen)
    gu0 = torch.randn(N, 2 * H, 2 * W, C, generator=gen)
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen) * 0.2
    drop0 = (torch.rand(N, C, generator=gen) > 0.3).float() / 0.7

    def run(device, stored):
        rt = Runtime(device, dtype)
        dev = rt.device
        def view(src, h, w):
            v = View.alloc(rt, N, h, w, Cp)
            v.dense()[..., :C] = src.to(dev, rt.tdtype)
            return v
        yv = view(y0, H, W)
        gd = view(gd0, H, W) if 'd' in srcs.split('+') else None
        gp = view(gp0, H // 2, W // 2) if 'pool' in srcs else None
        gu = view(gu0, 2 * H, 2 * W) if 'up' in srcs else None
        dm = None
        if 'drop' in srcs:
            dm = torch.ones(N, Cp, device=dev)
            dm[:, :C] = drop0.to(dev)
        stats = torch.zeros(16, 2, Cp, dtype=torch.float64, device=dev)
        yy = yv.dense().double()
        stats[0, 0] = yy.sum((0, 1, 2))
        stats[0, 1] = (yy * yy).sum((0, 1, 2))
        coef = rt.zeros((4, Cp), torch.float32)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        g_, b_ = gamma.to(dev), beta.to(dev)
        nv.call('segnb_bn_finalize', nv.ptr(stats), C, Cp, float(N * H * W), nv.ptr(g_), nv.ptr(b_), 1e-5, 0.1,
                nv.ptr(rm), nv.ptr(rv), nv.ptr(nbt), 1, nv.ptr(coef), rt.stream)
        stats.fill_(3.0)                       # (the forward statistics a fused apply clears)
        from segnb.engine import vld, vptr
        dz = View.alloc(rt, N, H, W, Cp)
        sums = rt.zeros((16, 2, Cp), torch.float64)
        nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm),
                vptr(gd), vld(gd), vptr(gp), vld(gp), vptr(gu), vld(gu), dz.ptr if stored else None, dz.ld, nv.ptr(sums), None, 0,
                rt.stream)
        bcoef = rt.zeros((3, Cp), torch.float32)
        dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dyv = View.alloc(rt, N, H, W, Cp)
        if stored:
            nv.call('segnb_bn_bwd_apply_fused', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums), nv.ptr(g_),
                    nv.ptr(bcoef), nv.ptr(dgam), nv.ptr(dbet), 0, nv.ptr(stats), dz.ptr, dz.ld, dyv.ptr, dyv.ld, rt.stream)
        else:
            nv.call('segnb_bn_bwd_apply_fused_src', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums),
                    nv.ptr(g_), nv.ptr(bcoef), nv.ptr(dgam), nv.ptr(dbet), 0, nv.ptr(stats), act, 0.01, nv.ptr(dm), vptr(gd),
                    vld(gd), vptr(gp), vld(gp), vptr(gu), vld(gu), dyv.ptr, dyv.ld, rt.stream)
        if device != 'cpu':
            torch.cuda.synchronize()
        assert float(stats.abs().max()) == 0.0
        return dyv.dense().float().cpu(), dgam.cpu(), dbet.cpu(), bcoef.cpu()

    a = run('cuda', True)
    b = run('cuda', False)
    for u, v in zip(a, b):
        if dtype == 'bf16':
            assert torch.equal(u, v)
        else:                              # (fp32: no storage rounding between the two passes; fused multiply-adds may differ)
            torch.testing.assert_close(u, v, rtol=2e-6, atol=2e-6 * float(v.abs().max()))
    with on_emulator():
        e = run('cpu', False)
    check('dy vs emulator', b[0], e[0], dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 20, 24, 32, 1, True), (3, 9, 7, 64, 3, False), (2, 16, 12, 16, 2, True), (1, 33, 40, 8, 4, True)],
                         ids=lambda s: 'N%d_%dx%d_C%d_K%d_%s' % (s[:5] + ('drop' if s[5] else 'nodrop',)))
def test_last_layer_and_classifier_in_one_pass(shape, dtype):
    """segnb_bn_fwd_fused_head == segnb_bn_fwd_fused + segnb_head_fwd and segnb_head_bn_bwd == segnb_head_bwd +
    segnb_bn_act_bwd_reduce (zf_unet.py:56-58,91-93 and their backward): the activated tensor and its gradient never go to
    memory.  dz, the activated values and the published coefficients are bit-equal (same expressions, same rounding points);
    logits / dw / db / sums differ by summation order only."""
    N, H, W, C, K, use_drop = shape
    rt = Runtime('cuda', dtype)
    Cp = cp.pad8(C)
    assert nv.query('segnb_head_fused_ok', K, Cp) == 1 and nv.query('segnb_head_fused_ok', 5, Cp) == 0
    gen = torch.Generator().manual_seed(31 * C + H + K)
    yv = _view_from(rt, torch.randn(N, H, W, C, generator=gen) * 1.5 + 0.3, Cp)
    gamma = (1 + 0.3 * torch.randn(C, generator=gen)).cuda()
    beta = (0.2 * torch.randn(C, generator=gen)).cuda()
    hw_ = (0.3 * torch.randn(K, C, generator=gen)).cuda()
    hb = (0.1 * torch.randn(K, generator=gen)).cuda()
    dl = torch.randn(N, K, H, W, generator=gen).cuda()
    dm = None
    if use_drop:
        dm = torch.ones(N, Cp, device='cuda')
        dm[:, :C] = ((torch.rand(N, C, generator=gen) > 0.3).float() / 0.7).cuda()
    act = nv.ACT_RELU
    res = {}
    for fused in (False, True):
        stats = rt.zeros((16, 2, Cp), torch.float64)
        nv.call('segnb_bn_stats', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(stats), rt.stream)
        sums = rt.zeros((16, 2, Cp), torch.float64)
        sums.fill_(7.0)                                   # the forward clears them
        coef = rt.zeros((4, Cp), torch.float32)
        rm, rvv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        nbt = torch.zeros((), dtype=torch.int64, device='cuda')
        a = View.alloc(rt, N, H, W, Cp)
        logits = torch.zeros(N, K, H, W, device='cuda')
        dz = View.alloc(rt, N, H, W, Cp)
        dw, db = torch.ones(K, C, device='cuda'), torch.ones(K, device='cuda')          # accumulate on top of 1
        head = (nv.ptr(stats), nv.ptr(gamma), nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm), nv.ptr(rvv), nv.ptr(nbt), nv.ptr(coef),
                nv.ptr(sums), act, 0.01, nv.ptr(dm))
        if fused:
            nv.call('segnb_bn_fwd_fused_head', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, *head, a.ptr, a.ld, nv.ptr(hw_),
                    nv.ptr(hb), K, nv.ptr(logits), rt.stream)
            logits2 = torch.zeros_like(logits)            # and without the activated tensor
            stats2, coef2 = stats.clone(), rt.zeros((4, Cp), torch.float32)
            rm2, rv2, nbt2 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
            nv.call('segnb_bn_fwd_fused_head', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(stats2), nv.ptr(gamma),
                    nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm2), nv.ptr(rv2), nv.ptr(nbt2), nv.ptr(coef2), None, act, 0.01, nv.ptr(dm),
                    None, 0, nv.ptr(hw_), nv.ptr(hb), K, nv.ptr(logits2), rt.stream)
            torch.cuda.synchronize()
            assert torch.equal(logits, logits2) and torch.equal(coef, coef2)
            assert float(sums.abs().max()) == 0.0
            nv.call('segnb_head_bn_bwd', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm),
                    nv.ptr(hw_), K, nv.ptr(dl), dz.ptr, dz.ld, nv.ptr(sums), nv.ptr(dw), nv.ptr(db), rt.stream)
        else:
            nv.call('segnb_bn_fwd_fused', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, *head, a.ptr, a.ld, None, 0, None, 0, None, 0,
                    rt.stream)
            nv.call('segnb_head_fwd', rt.code, a.ptr, a.ld, N, H, W, C, nv.ptr(hw_), nv.ptr(hb), K, nv.ptr(logits), rt.stream)
            da = View.alloc(rt, N, H, W, Cp)
            nv.call('segnb_head_bwd', rt.code, a.ptr, a.ld, N, H, W, C, Cp, nv.ptr(hw_), K, nv.ptr(dl), da.ptr, da.ld,
                    nv.ptr(dw), nv.ptr(db), rt.stream)
            nv.call('segnb_bn_act_bwd_reduce', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm),
                    da.ptr, da.ld, None, 0, None, 0, dz.ptr, dz.ld, nv.ptr(sums), None, 0, rt.stream)
        torch.cuda.synchronize()
        res[fused] = dict(a=a.t.clone(), logits=logits.clone(), coef=coef.clone(), rm=rm.clone(), rv=rvv.clone(), nbt=int(nbt),
                          dz=dz.t.clone(), sums=sums.sum(0).clone(), dw=dw.clone(), db=db.clone())
    u, f = res[False], res[True]
    assert torch.equal(u['a'], f['a']) and torch.equal(u['coef'], f['coef']) and torch.equal(u['dz'], f['dz'])
    assert torch.equal(u['rm'], f['rm']) and torch.equal(u['rv'], f['rv']) and u['nbt'] == f['nbt'] == 1
    torch.testing.assert_close(f['logits'], u['logits'], rtol=1e-5, atol=1e-5 * float(u['logits'].abs().max()))
    torch.testing.assert_close(f['dw'], u['dw'], rtol=1e-4, atol=1e-4 * float(u['dw'].abs().max()))
    torch.testing.assert_close(f['db'], u['db'], rtol=1e-4, atol=1e-4 * float(u['db'].abs().max()))
    torch.testing.assert_close(f['sums'], u['sums'], rtol=1e-6, atol=1e-6 * float(u['sums'].abs().max()))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 20, 24, 32, 1, True), (3, 9, 7, 64, 3, False), (2, 16, 12, 16, 2, True), (1, 33, 40, 8, 4, True),
                                   (32, 224, 224, 32, 1, True)],
                         ids=lambda s: 'N%d_%dx%d_C%d_K%d_%s' % (s[:5] + ('drop' if s[5] else 'nodrop',)))
def test_last_layer_dz_never_stored(shape, dtype):
    """segnb_head_bn_bwd(dz = NULL) + segnb_head_bn_bwd_apply == segnb_head_bn_bwd(dz) + segnb_bn_bwd_apply_fused, bit for bit (dy,
    the published coefficients, dgamma / dbeta, the cleared forward statistics): the last layer's dz (zf_unet.py:56-58,91-93) is a
    function of d(logits) and y and is recomputed by the apply pass instead of being written and read back.  Oracle: torch autograd
    of the same three modules (training-mode batch_norm on y -> ReLU -> Dropout2d multiplier -> 1 x 1 classifier) on the CPU; incl.
    the timed configuration's 32 x 224 x 224 x 32."""
    N, H, W, C, K, use_drop = shape
    if dtype == 'f32' and N * H * W > 100000:
        pytest.skip('the timed shape runs in the timed precision')
    rt = Runtime('cuda', dtype)
    Cp = cp.pad8(C)
    gen = torch.Generator().manual_seed(17 * C + H + K)
    y_host = torch.randn(N, H, W, C, generator=gen) * 1.5 + 0.3
    yv = _view_from(rt, y_host, Cp)
    gamma = (1 + 0.3 * torch.randn(C, generator=gen)).cuda()
    beta = (0.2 * torch.randn(C, generator=gen)).cuda()
    hw_ = (0.3 * torch.randn(K, C, generator=gen)).cuda()
    dl = (torch.randn(N, K, H, W, generator=gen) / (N * H * W) ** 0.5).cuda()
    dm = None
    if use_drop:
        dm = torch.ones(N, Cp, device='cuda')
        dm[:, :C] = ((torch.rand(N, C, generator=gen) > 0.3).float() / 0.7).cuda()
    act = nv.ACT_RELU
    out = {}
    for recompute in (False, True):
        stats = rt.zeros((16, 2, Cp), torch.float64)
        nv.call('segnb_bn_stats', rt.code, yv.ptr, yv.ld, N, H, W, Cp, nv.ptr(stats), rt.stream)
        keep_stats = stats.clone()
        sums, coef, bcoef = rt.zeros((16, 2, Cp), torch.float64), rt.zeros((4, Cp), torch.float32), rt.zeros((3, Cp), torch.float32)
        rm, rvv, nbt = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
        nv.call('segnb_bn_finalize_keep', nv.ptr(stats), C, Cp, float(N * H * W), nv.ptr(gamma), nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm),
                nv.ptr(rvv), nv.ptr(nbt), nv.ptr(coef), nv.ptr(sums), rt.stream)
        dz = View.alloc(rt, N, H, W, Cp)
        dz.t.fill_(3.0)
        dw, db = torch.zeros(K, C, device='cuda'), torch.zeros(K, device='cuda')
        dgam, dbet = torch.ones(C, device='cuda'), torch.ones(C, device='cuda')          # accumulate on top of 1
        nv.call('segnb_head_bn_bwd', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), act, 0.01, nv.ptr(dm), nv.ptr(hw_), K,
                nv.ptr(dl), None if recompute else dz.ptr, dz.ld, nv.ptr(sums), nv.ptr(dw), nv.ptr(db), rt.stream)
        if recompute:
            torch.cuda.synchronize()
            assert float((dz.t.float() - 3.0).abs().max()) == 0.0, 'dz was written'
            nv.call('segnb_head_bn_bwd_apply', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums), nv.ptr(gamma),
                    nv.ptr(bcoef), nv.ptr(dgam), nv.ptr(dbet), 1, nv.ptr(stats), act, 0.01, nv.ptr(dm), nv.ptr(hw_), K, nv.ptr(dl),
                    dz.ptr, dz.ld, rt.stream)
        else:
            nv.call('segnb_bn_bwd_apply_fused', rt.code, yv.ptr, yv.ld, N, H, W, C, Cp, nv.ptr(coef), nv.ptr(sums), nv.ptr(gamma),
                    nv.ptr(bcoef), nv.ptr(dgam), nv.ptr(dbet), 1, nv.ptr(stats), dz.ptr, dz.ld, dz.ptr, dz.ld, rt.stream)
        torch.cuda.synchronize()
        assert float(keep_stats.abs().max()) > 0 and float(stats.abs().max()) == 0.0       # forward statistics cleared
        out[recompute] = dict(dy=dz.t.clone(), bcoef=bcoef.clone(), dgam=dgam.clone(), dbet=dbet.clone(), dw=dw.clone(), db=db.clone())
    u, r = out[False], out[True]
    for k in ('dy', 'bcoef', 'dgam', 'dbet', 'dw', 'db'):
        assert torch.equal(u[k], r[k]), k
    # oracle: torch autograd of batch_norm(training) -> relu -> dropout multiplier -> 1x1 classifier on the CPU
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    yr = yv.dense()[..., :C].float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    g_t, b_t = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    a = torch.relu(F.batch_norm(yr, None, None, g_t, b_t, True, 0.1, 1e-5))
    if dm is not None:
        a = a * dm[:, :C].cpu()[:, :, None, None]
    logits = torch.einsum('nchw,kc->nkhw', a, hw_.cpu())
    logits.backward(dl.cpu())
    dy_t = yr.grad.permute(0, 2, 3, 1)
    got = View(r['dy'], N, H, W, Cp).dense()[..., :C].float().cpu()
    scale = float(dy_t.abs().max())
    tolr = 3e-2 if dtype == 'bf16' else 2e-4             # (bf16: a, da and dz are each rounded to the storage type on the way)
    assert float((got - dy_t).abs().max()) <= tolr * scale, float((got - dy_t).abs().max()) / scale
    np.testing.assert_allclose((r['dgam'] - 1).cpu().numpy(), g_t.grad.numpy(), rtol=2e-2 if dtype == 'bf16' else 1e-3,
                               atol=(2e-2 if dtype == 'bf16' else 1e-3) * float(g_t.grad.abs().max()))
    np.testing.assert_allclose((r['dbet'] - 1).cpu().numpy(), b_t.grad.numpy(), rtol=2e-2 if dtype == 'bf16' else 1e-3,
                               atol=(2e-2 if dtype == 'bf16' else 1e-3) * float(b_t.grad.abs().max()))


@pytest.mark.parametrize('dtype', DTYPES)
def test_prefix_statistics_table(dtype):
    """tiramisu.py:9-44: a DenseLayer's BatchNorm covers the whole concat prefix.  The statistics of the prefix are those of its
    slices: segnb_bn_act_fwd_stats sums the slice it writes into its range of the buffer's table, segnb_bn_stats_ld sums a slice
    somebody else wrote, segnb_bn_fwd_fused_ld reads a prefix range -- together == segnb_bn_stats over the prefix +
    segnb_bn_fwd_fused (coefficients to fp64 summation order, the activated tensor to the storage rounding)."""
    rt = Runtime('cuda', dtype)
    N, H, W = 2, 12, 10
    widths = [24, 16, 16]                      # block input + two growth slices, all multiples of 8
    Cb = sum(widths)
    gen = torch.Generator().manual_seed(77)
    buf = View.alloc(rt, N, H, W, Cb)
    buf.t.zero_()
    table = rt.zeros((16, 2, Cb), torch.float64)
    off = 0
    dm = torch.ones(N, 16, device='cuda')
    dm[:, :] = ((torch.rand(N, 16, generator=gen) > 0.3).float() / 0.7).cuda()
    for k, wd in enumerate(widths):
        src = _view_from(rt, torch.randn(N, H, W, wd, generator=gen) * (1 + k) + 0.1 * k, wd)
        sl = buf.slice(off, wd)
        if k == 0:             # written by something else (a convolution epilogue): summed by a pass of its own
            nv.call('segnb_add', rt.code, None, 0, src.ptr, src.ld, sl.ptr, sl.ld, N, H, W, wd, rt.stream)
            nv.call('segnb_bn_stats_ld', rt.code, sl.ptr, sl.ld, N, H, W, wd, nv.ptr(table, off), Cb, rt.stream)
        else:                  # written by the Dropout2d pass of the layer's convolution: that pass sums it
            nv.call('segnb_bn_act_fwd_stats', rt.code, src.ptr, src.ld, N, H, W, wd, None, nv.ACT_NONE, 0.0, nv.ptr(dm),
                    sl.ptr, sl.ld, nv.ptr(table, off), Cb, rt.stream)
        off += wd
    # a layer over the first two slices (prefix of 40 channels), one over all three
    for wl in (widths[0] + widths[1], Cb):
        pre = buf.slice(0, wl)
        gamma = (1 + 0.2 * torch.randn(wl, generator=gen)).cuda()
        beta = (0.1 * torch.randn(wl, generator=gen)).cuda()
        outs = []
        for cached in (False, True):
            coef = rt.zeros((4, wl), torch.float32)
            sums = rt.zeros((16, 2, wl), torch.float64)
            rm, rvv = torch.zeros(wl, device='cuda'), torch.ones(wl, device='cuda')
            nbt = torch.zeros((), dtype=torch.int64, device='cuda')
            a = View.alloc(rt, N, H, W, wl)
            if cached:
                before = table.clone()
                nv.call('segnb_bn_fwd_fused_ld', rt.code, pre.ptr, pre.ld, N, H, W, wl, wl, nv.ptr(table), Cb, nv.ptr(gamma),
                        nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm), nv.ptr(rvv), nv.ptr(nbt), nv.ptr(coef), nv.ptr(sums), nv.ACT_RELU,
                        0.0, None, a.ptr, a.ld, rt.stream)
                torch.cuda.synchronize()
                assert torch.equal(table, before)            # read-only: the next layer reads it again
            else:
                stats = rt.zeros((16, 2, wl), torch.float64)
                nv.call('segnb_bn_stats', rt.code, pre.ptr, pre.ld, N, H, W, wl, nv.ptr(stats), rt.stream)
                nv.call('segnb_bn_fwd_fused', rt.code, pre.ptr, pre.ld, N, H, W, wl, wl, nv.ptr(stats), nv.ptr(gamma),
                        nv.ptr(beta), 1e-5, 0.1, nv.ptr(rm), nv.ptr(rvv), nv.ptr(nbt), nv.ptr(coef), nv.ptr(sums), nv.ACT_RELU,
                        0.0, None, a.ptr, a.ld, None, 0, None, 0, None, 0, rt.stream)
            torch.cuda.synchronize()
            outs.append((coef.clone(), a.t.clone().float(), rm.clone(), rvv.clone()))
        (c0, a0, m0, v0), (c1, a1, m1, v1) = outs
        torch.testing.assert_close(c1, c0, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m1, m0, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(v1, v0, rtol=1e-5, atol=1e-7)
        check('activated prefix', a1, a0, dtype)
