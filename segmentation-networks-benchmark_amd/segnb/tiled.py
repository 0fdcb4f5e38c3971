"""Tiled inference with D4 test-time augmentation on the device -- the data flow of the reference's
``inria_submit.predict_tiled`` (/root/reference/inria_submit.py:237-257):

    slicer = ImageSlicer(image.shape, patch_size, patch_size // 2, weight='pyramid')
    patches = tta_d4_aug(slicer.split(image)); batches -> model -> sigmoid -> tta_d4_deaug -> slicer.merge

The reference builds 8x the tiles on the host, ships every batch over PCIe and back and merges in numpy float64; here
the normalised image is uploaded once, every batch is gathered on the GPU straight into the model's NCHW input
(reflect-101 padding and the D4 transform are index maps of that gather), the logits of all items stay in HBM and ONE
kernel undoes the transforms, averages, weights and normalises (gather form: no atomics, the reference's summation
order)."""
import numpy as np
import torch

from lib.tiles import ImageSlicer
from . import _native as nv


def predict_tiled(image, model, patch_size, batch_size, test_transform=None, weight='pyramid'):
    """image: HxWxC array (already normalised unless `test_transform` is given: then `image, _ = test_transform(image)`
    as in the reference).  model: a segnb-backed module on a GPU, called in eval mode without autograd.  Returns the
    float32 HxW (K=1) or HxWxK probability mask."""
    if test_transform is not None:
        image, _ = test_transform(image)
    image = np.ascontiguousarray(image, dtype=np.float32)
    if image.ndim == 2:
        image = image[..., None]
    H, W, C = image.shape
    slicer = ImageSlicer(image.shape, patch_size, patch_size // 2, weight=weight)
    device = next(model.parameters()).device
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0
    img = torch.from_numpy(image).to(device)
    crops = torch.tensor([[c[0], c[1]] for c in slicer.crops], dtype=torch.int32, device=device)
    wt = torch.from_numpy(np.ascontiguousarray(slicer.compute_weight(patch_size), dtype=np.float64)).to(device)
    ntiles, S = len(slicer.crops), patch_size
    nx, ny = slicer.grid()
    nitems = ntiles * 8
    was_training = model.training
    model.eval()
    logits, K = None, None
    with torch.no_grad():
        for first in range(0, nitems, batch_size):
            count = min(batch_size, nitems - first)
            x = torch.empty((count, C, S, S), dtype=torch.float32, device=device)
            nv.call('segnb_tiles_gather', nv.ptr(img), H, W, C, slicer.margin_top, slicer.margin_left, nv.ptr(crops),
                    first, count, S, nv.ptr(x), stream)
            y = model(x)
            if logits is None:
                K = y.shape[1]
                logits = torch.empty((nitems, K, S, S), dtype=torch.float32, device=device)
            logits[first:first + count] = y
        out = torch.empty((H, W, K), dtype=torch.float32, device=device)
        nv.call('segnb_tiles_merge', nv.ptr(logits), K, S, nv.ptr(crops), ntiles, slicer.tile_step, nx, ny, nv.ptr(wt),
                H, W, slicer.margin_top, slicer.margin_left, nv.ptr(out), stream)
    if was_training:
        model.train()
    return out.cpu().numpy()
