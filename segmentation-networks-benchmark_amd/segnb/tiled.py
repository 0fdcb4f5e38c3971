"""Whole-image inference with D4 test-time augmentation on the device -- the data flow of the reference's
``inria_submit.predict_tiled`` / ``predict_full`` (/root/reference/inria_submit.py:217-257):

    slicer = ImageSlicer(image.shape, patch_size, patch_size // 2, weight='pyramid')
    patches = tta_d4_aug(slicer.split(image)); batches -> model -> sigmoid -> tta_d4_deaug -> slicer.merge

The reference builds 8x the tiles on the host, ships every batch over PCIe and back and merges in numpy float64; here
the normalised image is uploaded once, every batch is gathered on the GPU straight into the model's NCHW input
(reflect-101 padding and the D4 transform are index maps of that gather), the logits of all items stay in HBM and ONE
kernel undoes the transforms, averages, weights and normalises (gather form: no atomics, the reference's summation
order).

Data parallel: the (tile, transform) item list is cut into one contiguous range per rank; every
rank runs its range and the ranges are all-gathered into the full logits buffer (no reduction: results are
bit-identical to one GPU); every rank then holds the merged mask."""
import numpy as np
import torch
import torch.distributed as td

from lib.tiles import ImageSlicer
from lib.augmentations import D4
from . import _native as nv


def _world():
    if td.is_available() and td.is_initialized():
        return td.get_world_size(), td.get_rank()
    return 1, 0


def item_range(nitems, world, rank):
    """Contiguous share [lo, hi) of rank `rank` and the common chunk length (the all-gather needs equal chunks)."""
    chunk = (nitems + world - 1) // world
    return min(rank * chunk, nitems), min((rank + 1) * chunk, nitems), chunk


def predict_tiled(image, model, test_transform, patch_size, batch_size, weight='pyramid'):
    """Same positional signature as inria_submit.predict_tiled (:237).  image: HxWxC array; ``test_transform`` is
    applied as ``image, _ = test_transform(image)`` (pass None for an already normalised image).  model: a
    segnb-backed module on a GPU, called in eval mode without autograd.  Returns the float32 HxW (K=1) or HxWxK
    probability mask."""
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
    nitems = ntiles * len(D4)
    world, rank = _world()
    lo, hi, chunk = item_range(nitems, world, rank)
    was_training = model.training
    model.eval()
    logits, K = None, None
    with torch.no_grad():
        # every forward runs at the full batch size (the short last batch is padded with stale items whose outputs are
        # dropped), so the model keeps ONE set of activation buffers
        x = torch.zeros((batch_size, C, S, S), dtype=torch.float32, device=device)
        for first in range(lo, max(hi, lo + 1), batch_size):
            count = min(batch_size, hi - first)
            if count > 0:
                nv.call('segnb_tiles_gather', nv.ptr(img), H, W, C, slicer.margin_top, slicer.margin_left,
                        nv.ptr(crops), first, count, S, nv.ptr(x), stream)
            y = model(x)
            if logits is None:
                K = y.shape[1]
                logits = torch.zeros((world * chunk, K, S, S), dtype=torch.float32, device=device)
            if count > 0:
                logits[first:first + count] = y[:count]
        if world > 1:
            parts = [logits[r * chunk:(r + 1) * chunk] for r in range(world)]
            td.all_gather(parts, parts[rank].clone())
        out = torch.empty((H, W, K), dtype=torch.float32, device=device)
        nv.call('segnb_tiles_merge', nv.ptr(logits), K, S, nv.ptr(crops), ntiles, slicer.tile_step, nx, ny, nv.ptr(wt),
                H, W, slicer.margin_top, slicer.margin_left, nv.ptr(out), stream)
    if was_training:
        model.train()
    return out.cpu().numpy()


def pad_to_multiple(image, pad_size):
    """lib/augmentations.py:513-532 with BORDER_REPLICATE (numpy 'edge'), including its quirk: when only ONE side is
    not a multiple of pad_size the other one still grows by a full pad_size."""
    rows, cols = image.shape[:2]
    pr, pc = rows % pad_size, cols % pad_size
    if pr == 0 and pc == 0:
        return image, (0, 0, 0, 0)
    pr, pc = pad_size - pr, pad_size - pc
    top, left = pr // 2, pc // 2
    pads = (top, pr - top, left, pc - left)
    spec = [(pads[0], pads[1]), (pads[2], pads[3])] + [(0, 0)] * (image.ndim - 2)
    return np.pad(image, spec, mode='edge'), pads


def predict_full(image, model, test_transform):
    """inria_submit.predict_full (:217-234): pad to a multiple of 32 (replicated border), normalise, run the 8 D4
    views of the WHOLE image, sigmoid, undo the views, average, unpad.  (The reference ends in ``next(list)``, a
    TypeError on current Python -- SURVEY 8f; the intended value, the single de-augmented mask, is returned.)  The
    views are formed on the device from one upload; accumulation runs in the table order of tta_d4_deaug."""
    image, pads = pad_to_multiple(np.asarray(image), 32)
    if test_transform is not None:
        image, _ = test_transform(image)
    image = np.ascontiguousarray(image, dtype=np.float32)
    if image.ndim == 2:
        image = image[..., None]
    device = next(model.parameters()).device
    x = torch.from_numpy(image).to(device).permute(2, 0, 1).unsqueeze(0)          # [1, C, H, W]
    was_training = model.training
    model.eval()
    acc = None
    with torch.no_grad():
        for k, mirrored in D4:
            v = torch.rot90(x, k, dims=(2, 3))
            if mirrored:
                v = torch.flip(v, dims=(3,))
            p = torch.sigmoid(model(v.contiguous()))
            if mirrored:
                p = torch.flip(p, dims=(3,))
            p = torch.rot90(p, -k, dims=(2, 3))
            acc = p if acc is None else acc + p
        mask = (acc * float(1. / 8.))[0].permute(1, 2, 0)                         # [H, W, K]
    if was_training:
        model.train()
    mask = mask.cpu().numpy()
    if mask.shape[2] == 1:
        mask = mask[..., 0]
    top, btm, left, right = pads
    return mask[top:mask.shape[0] - btm, left:mask.shape[1] - right]
