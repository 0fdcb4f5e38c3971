"""CPU restatement of the tiled-inference helpers.  TEST INFRASTRUCTURE (tests/, smoke(), bench cpu_baseline only).

  compute_patch_weight_loss, ImageSlicer   lib/tiles.py:6-27, 30-168
  tta_d4_aug / tta_d4_deaug                lib/augmentations.py:476-511
  predict_tiled (the data flow)            inria_submit.py:237-257

Pinned against the reference (tests/golden/tiles.npz, made by tests/golden/make_golden.py importing lib/tiles.py):
the pyramid weights, margins / crops of several image shapes and merge(); and (round 3) against
tests/golden/augment.npz -- the reference's own split() / cut_patch() (lib/tiles.py:98-135) and tta_d4_aug /
tta_d4_deaug (lib/augmentations.py:476-511) run on non-symmetric arrays by make_golden.py gen_augment: tiles, patches and
the D4 pair bit for bit.  One dependency stays restated there: cv2 is absent from the image, so the generator gives the
reference's code a copyMakeBorder that is numpy.pad(mode='reflect') (BORDER_REFLECT101: mirror without repeating the edge
pixel -- SURVEY 8c); what is pinned is the reference's margin / crop / indexing logic around it.
"""
import math

import numpy as np


def compute_patch_weight_loss(width, height):
    """tiles.py:6-27 -- W = alpha * De / (Dc + De), Dc = distance to the tile centre, De = distance to the nearest
    edge (each measured at half-pixel offsets exactly as the reference's double loop does)."""
    i = np.arange(width, dtype=np.float64)[:, None]
    j = np.arange(height, dtype=np.float64)[None, :]
    Dc = np.sqrt(np.square(i - width * 0.5 + 0.5) + np.square(j - height * 0.5 + 0.5))
    half = np.float64(0.25)                                    # square(x - x + 0.5)
    De_l = np.sqrt(np.square(i - 0 + 0.5) + half) + 0 * j
    De_r = np.sqrt(np.square(i - width + 0.5) + half) + 0 * j
    De_b = np.sqrt(half + np.square(j - 0 + 0.5)) + 0 * i
    De_t = np.sqrt(half + np.square(j - height + 0.5)) + 0 * i
    De = np.minimum(np.minimum(De_l, De_r), np.minimum(De_b, De_t))
    alpha = (width * height) / np.sum(np.divide(De, np.add(Dc, De)))
    W = alpha * np.divide(De, np.add(Dc, De))
    return W, Dc, De


class ImageSlicer(object):
    """tiles.py:30-168, same attributes and methods."""

    def __init__(self, image_shape, tile_size, tile_step=0, image_margin=0, weight='mean'):
        self.image_height = image_shape[0]
        self.image_width = image_shape[1]
        self.tile_size = tile_size
        self.tile_step = tile_step
        self.compute_weight = {'mean': self._mean, 'pyramid': self._pyramid}[weight]
        if tile_step < 1 or tile_step > tile_size:
            raise ValueError()
        overlap = tile_size - tile_step
        self.margin_left = self.margin_right = self.margin_top = self.margin_bottom = 0
        if image_margin == 0:
            nw = max(1, math.ceil((self.image_width - overlap) / tile_step))
            nh = max(1, math.ceil((self.image_height - overlap) / tile_step))
            extra_w = self.tile_step * nw - (self.image_width - overlap)
            extra_h = self.tile_step * nh - (self.image_height - overlap)
            self.margin_left = extra_w // 2
            self.margin_right = extra_w - self.margin_left
            self.margin_top = extra_h // 2
            self.margin_bottom = extra_h - self.margin_top
        else:
            if (self.image_width - overlap + 2 * image_margin) % tile_step != 0:
                raise ValueError()
            if (self.image_height - overlap + 2 * image_margin) % tile_step != 0:
                raise ValueError()
            self.margin_left = self.margin_right = self.margin_top = self.margin_bottom = image_margin
        self.crops = []
        for y in range(0, self.image_height + self.margin_top + self.margin_bottom - tile_size + 1, tile_step):
            for x in range(0, self.image_width + self.margin_left + self.margin_right - tile_size + 1, tile_step):
                self.crops.append((x, y, tile_size, tile_size))

    def _padded(self, image):
        assert image.shape[0] == self.image_height and image.shape[1] == self.image_width
        pad = [(self.margin_top, self.margin_bottom), (self.margin_left, self.margin_right)] + \
              [(0, 0)] * (image.ndim - 2)
        return np.pad(image, pad, mode='reflect')

    def split(self, image):
        image = self._padded(image)
        return [image[y:y + th, x:x + tw].copy() for x, y, tw, th in self.crops]

    def cut_patch(self, image, slice_index):
        x, y, tw, th = self.crops[slice_index]
        return self._padded(image)[y:y + th, x:x + tw].copy()

    def merge(self, tiles, dtype=np.float32):
        if len(tiles) != len(self.crops):
            raise ValueError
        channels = 1 if len(tiles[0].shape) == 2 else tiles[0].shape[2]
        shape = (self.image_height + self.margin_bottom + self.margin_top,
                 self.image_width + self.margin_right + self.margin_left, channels)
        image = np.zeros(shape, dtype=np.float64)
        norm = np.zeros(shape, dtype=np.float64)
        w = np.dstack([self.compute_weight(self.tile_size)] * channels)
        for tile, (x, y, tw, th) in zip(tiles, self.crops):
            image[y:y + th, x:x + tw] += tile.reshape(th, tw, channels) * w
            norm[y:y + th, x:x + tw] += w
        norm = np.clip(norm, a_min=np.finfo(norm.dtype).eps, a_max=None)
        out = np.divide(image, norm).astype(dtype)
        return out[self.margin_top:self.image_height + self.margin_top,
                   self.margin_left:self.image_width + self.margin_left]

    def _mean(self, tile_size):
        return np.ones((tile_size, tile_size), dtype=np.float32)

    def _pyramid(self, tile_size):
        return compute_patch_weight_loss(tile_size, tile_size)[0]


def tta_d4_aug(images):
    """augmentations.py:476-491: the 8 elements of D4 per image, in the reference's order."""
    res = []
    for image in images:
        res.extend([image, np.rot90(image, 1), np.rot90(image, 2), np.rot90(image, 3), np.fliplr(image),
                    np.fliplr(np.rot90(image, 1)), np.fliplr(np.rot90(image, 2)), np.fliplr(np.rot90(image, 3))])
    return res


def tta_d4_deaug(image_list):
    """augmentations.py:494-511: undo each transform, average the 8."""
    assert len(image_list) % 8 == 0
    res = []
    for i in range(0, len(image_list), 8):
        g = image_list[i:i + 8]
        img = (g[0] + np.rot90(g[1], -1) + np.rot90(g[2], -2) + np.rot90(g[3], -3) + np.fliplr(g[4]) +
               np.rot90(np.fliplr(g[5]), -1) + np.rot90(np.fliplr(g[6]), -2) + np.rot90(np.fliplr(g[7]), -3)) * float(1. / 8.)
        res.append(img)
    return res


def predict_tiled(image, logits_fn, patch_size, batch_size):
    """inria_submit.py:237-257 with `logits_fn(batch NCHW float32 ndarray) -> logits [B,1,S,S]` in place of the
    model call; image is the already normalised HxWxC float array."""
    slicer = ImageSlicer(image.shape, patch_size, patch_size // 2, weight='pyramid')
    patches = tta_d4_aug(slicer.split(image))
    preds = []
    for s in range(0, len(patches), batch_size):
        x = np.stack([np.moveaxis(p, -1, 0) for p in patches[s:s + batch_size]]).astype(np.float32)
        y = 1.0 / (1.0 + np.exp(-logits_fn(x).astype(np.float64)))
        preds.extend(np.moveaxis(y, 1, -1).astype(np.float32))
    return slicer.merge(tta_d4_deaug(preds), dtype=np.float32)


def pad(image, pad_size):
    """augmentations.py:513-532 with cv2.BORDER_REPLICATE == numpy 'edge' (inria_submit.py:218).  Quirk kept: when
    only one side is not a multiple of pad_size, the other side is still padded by a full pad_size."""
    rows, cols = image.shape[:2]
    pad_rows, pad_cols = rows % pad_size, cols % pad_size
    if pad_rows == 0 and pad_cols == 0:
        return image, (0, 0, 0, 0)
    pad_rows, pad_cols = pad_size - pad_rows, pad_size - pad_cols
    pad_top = pad_rows // 2
    pad_btm = pad_rows - pad_top
    pad_left = pad_cols // 2
    pad_right = pad_cols - pad_left
    widths = [(pad_top, pad_btm), (pad_left, pad_right)] + [(0, 0)] * (image.ndim - 2)
    return np.pad(image, widths, mode='edge'), (pad_top, pad_btm, pad_left, pad_right)


def unpad(image, pads):
    """augmentations.py:535-538"""
    pad_top, pad_btm, pad_left, pad_right = pads
    rows, cols = image.shape[:2]
    return image[pad_top:rows - pad_btm, pad_left:cols - pad_right]


def predict_full(image, logits_fn):
    """inria_submit.py:217-234 with `logits_fn(batch NCHW float32 ndarray) -> logits [1,K,h,w]` in place of the model
    call and an identity test_transform.  The reference's last statement, ``next(aug.tta_d4_deaug(predicts))``, raises
    TypeError on a list; the evident intent -- the one de-augmented mask -- is what is returned here."""
    image, pads = pad(image, 32)
    predicts = []
    for view in tta_d4_aug([image]):
        x = np.ascontiguousarray(np.moveaxis(view, -1, 0))[None].astype(np.float32)
        y = 1.0 / (1.0 + np.exp(-logits_fn(x).astype(np.float64)))
        predicts.append(np.squeeze(np.moveaxis(y.astype(np.float32), 1, -1)))
    return unpad(tta_d4_deaug(predicts)[0], pads)
