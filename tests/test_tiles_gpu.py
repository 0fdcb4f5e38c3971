"""-m gpu: tiled inference with D4 TTA on the HIP kernels (segnb_tiles_gather / segnb_tiles_merge) vs the oracle's
restatement of inria_submit.predict_tiled.  Tolerance: sigmoid differs by an fp32 ulp between implementations ->
2e-6 on probabilities."""
import time

import numpy as np
import pytest
import torch

from oracle import tiles_ref

pytestmark = pytest.mark.gpu


class _Lin(torch.nn.Module):
    def __init__(self, S, C):
        super().__init__()
        self.w = torch.nn.Parameter(torch.linspace(-1.1, 0.7, C))
        self.register_buffer('ramp', torch.linspace(-1, 1, S)[None, None, :, None] * 0.5 +
                             torch.linspace(-0.3, 0.6, S)[None, None, None, :])

    def forward(self, x):
        y = (x * self.w[None, :, None, None]).sum(1, keepdim=True) + self.ramp
        return torch.cat([y, -0.5 * y + 0.1], 1)              # two classes: exercises K > 1


@pytest.mark.parametrize('shape,S', [((50, 71, 3), 16), ((33, 32, 3), 16), ((130, 97, 1), 32), ((64, 64, 4), 64)])
def test_gather_merge_kernels_vs_oracle(shape, S):
    from segnb.tiled import predict_tiled
    rng = np.random.RandomState(3)
    img = rng.randn(*shape).astype(np.float32)
    model = _Lin(S, shape[2]).cuda()
    got = predict_tiled(img, model, None, S, 7)
    cpu = _Lin(S, shape[2])
    with torch.no_grad():
        ref = tiles_ref.predict_tiled(img, lambda x: cpu(torch.from_numpy(x)).numpy(), S, 7)
    assert got.shape == (shape[0], shape[1], 2)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)


def test_gather_kernel_reproduces_reference_split(golden_dir):
    """segnb_tiles_gather, transform 0, vs the tiles the reference's ImageSlicer.split produced (tests/golden/augment.npz:
    lib/tiles.py:98-120 run by make_golden.py) -- incl. the margin case and single-channel images; bitwise (a gather)."""
    import os
    from lib.tiles import ImageSlicer
    from segnb import _native as nv
    g = np.load(os.path.join(golden_dir, 'augment.npz'))
    k = 0
    while 'split%d/args' % k in g.files:
        a = g['split%d/args' % k]
        shape, ts, step, margin = tuple(int(v) for v in a[:int(a[3])]), int(a[4]), int(a[5]), int(a[6])
        image = g['split%d/image' % k]
        img3 = image if image.ndim == 3 else image[..., None]
        H, W, C = img3.shape
        sl = ImageSlicer(shape, ts, step, margin)
        img = torch.from_numpy(np.ascontiguousarray(img3)).cuda()
        crops = torch.tensor([[c[0], c[1]] for c in sl.crops], dtype=torch.int32).cuda()
        want = g['split%d/tiles' % k]
        n = len(sl.crops)
        x = torch.zeros((8 * n, C, ts, ts), dtype=torch.float32).cuda()
        nv.call('segnb_tiles_gather', nv.ptr(img), H, W, C, sl.margin_top, sl.margin_left, nv.ptr(crops), 0, 8 * n, ts,
                nv.ptr(x), torch.cuda.current_stream().cuda_stream)
        got = x.view(n, 8, C, ts, ts)[:, 0].permute(0, 2, 3, 1).cpu().numpy()
        assert np.array_equal(got.reshape(want.shape), want), k
        k += 1
    assert k == 4


def test_predict_tiled_zf_unet_eval_and_throughput():
    """the real model in the loop (eval mode, bf16 kernels): device flow == oracle flow fed by the same model, and a
    throughput figure for the record (1024x1024 image, 256-pixel tiles, 49 tiles x 8 transforms)."""
    from lib.models.zf_unet import ZF_UNET
    from segnb.tiled import predict_tiled
    torch.manual_seed(0)
    model = ZF_UNET(filters=8).cuda().eval()
    rng = np.random.RandomState(4)
    img = rng.randn(200, 300, 3).astype(np.float32)
    got = predict_tiled(img, model, None, 64, 16)

    def logits_fn(x):
        with torch.no_grad():
            return model(torch.from_numpy(x).cuda()).float().cpu().numpy()
    ref = tiles_ref.predict_tiled(img, logits_fn, 64, 16)
    np.testing.assert_allclose(got[..., 0], ref[..., 0] if ref.ndim == 3 else ref, rtol=1e-5, atol=1e-5)
    big = rng.randn(1024, 1024, 3).astype(np.float32)
    full = ZF_UNET().cuda().eval()
    predict_tiled(big, full, None, 256, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mask = predict_tiled(big, full, None, 256, 32)
    dt = time.perf_counter() - t0
    assert mask.shape == (1024, 1024, 1) and np.isfinite(mask).all() and 0.0 <= mask.min() and mask.max() <= 1.0
    print('predict_tiled 1024x1024, 256-px tiles, D4 TTA, ZF_UNET bf16: %.1f ms (%.0f tile-forwards/s)'
          % (dt * 1e3, 49 * 8 / dt))
