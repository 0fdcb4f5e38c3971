"""Tile slicing / weighted merging -- drop-in for the reference's ``lib.tiles`` (/root/reference/lib/tiles.py:6-168):
``compute_patch_weight_loss`` and ``ImageSlicer`` with the same constructor, attributes (``crops``, ``margin_*``,
``tile_size`` ...), methods and results.  The numpy methods are host code like the reference's (vectorised: the
pyramid weight was an O(w*h) Python double loop); the device path used by ``segnb.tiled.predict_tiled`` keeps
images, tiles and the accumulators on the GPU (segnb_tiles_gather / segnb_tiles_merge).

cv2 is not a dependency: BORDER_REFLECT101 of ``cv2.copyMakeBorder`` is ``numpy.pad(mode='reflect')``; other
border types are rejected."""
import math

import numpy as np

BORDER_REFLECT101 = 4       # the value of cv2.BORDER_REFLECT101, accepted for signature compatibility


def compute_patch_weight_loss(width, height):
    """W = alpha * De / (Dc + De): Dc distance to the tile centre, De to the nearest edge (tiles.py:6-27)."""
    i = np.arange(width, dtype=np.float64)[:, None]
    j = np.arange(height, dtype=np.float64)[None, :]
    Dc = np.sqrt(np.square(i - width * 0.5 + 0.5) + np.square(j - height * 0.5 + 0.5))
    q = np.float64(0.25)
    De = np.minimum(np.minimum(np.sqrt(np.square(i + 0.5) + q) + 0 * j, np.sqrt(np.square(i - width + 0.5) + q) + 0 * j),
                    np.minimum(np.sqrt(q + np.square(j + 0.5)) + 0 * i, np.sqrt(q + np.square(j - height + 0.5)) + 0 * i))
    ratio = np.divide(De, np.add(Dc, De))
    alpha = (width * height) / np.sum(ratio)
    return alpha * ratio, Dc, De


class ImageSlicer:
    """Helper class to slice image into tiles and merge them back with fusion (tiles.py:30-168)."""

    def __init__(self, image_shape, tile_size, tile_step=0, image_margin=0, weight='mean'):
        self.image_height = image_shape[0]
        self.image_width = image_shape[1]
        self.tile_size = tile_size
        self.tile_step = tile_step
        weights = {'mean': self._mean, 'pyramid': self._pyramid}
        self.compute_weight = weights[weight]
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

    # ---- host (numpy) API of the reference ---------------------------------------------------------------
    def _padded(self, image, borderType):
        assert image.shape[0] == self.image_height
        assert image.shape[1] == self.image_width
        if borderType != BORDER_REFLECT101:
            raise ValueError('only BORDER_REFLECT101 is supported')
        pad = [(self.margin_top, self.margin_bottom), (self.margin_left, self.margin_right)] + \
              [(0, 0)] * (image.ndim - 2)
        return np.pad(image, pad, mode='reflect')

    def split(self, image, borderType=BORDER_REFLECT101, value=0):
        image = self._padded(image, borderType)
        return [image[y:y + th, x:x + tw].copy() for x, y, tw, th in self.crops]

    def cut_patch(self, image, slice_index, borderType=BORDER_REFLECT101, value=0):
        x, y, tw, th = self.crops[slice_index]
        return self._padded(image, borderType)[y:y + th, x:x + tw].copy()

    def merge(self, tiles, dtype=np.float32):
        if len(tiles) != len(self.crops):
            raise ValueError
        channels = 1 if len(tiles[0].shape) == 2 else tiles[0].shape[2]
        target_shape = (self.image_height + self.margin_bottom + self.margin_top,
                        self.image_width + self.margin_right + self.margin_left, channels)
        image = np.zeros(target_shape, dtype=np.float64)
        norm_mask = np.zeros(target_shape, dtype=np.float64)
        w = np.dstack([self.compute_weight(self.tile_size)] * channels)
        for tile, (x, y, tw, th) in zip(tiles, self.crops):
            image[y:y + th, x:x + tw] += tile.reshape(th, tw, channels) * w
            norm_mask[y:y + th, x:x + tw] += w
        norm_mask = np.clip(norm_mask, a_min=np.finfo(norm_mask.dtype).eps, a_max=None)
        normalized = np.divide(image, norm_mask).astype(dtype)
        crop = normalized[self.margin_top:self.image_height + self.margin_top,
                          self.margin_left:self.image_width + self.margin_left]
        assert crop.shape[0] == self.image_height
        assert crop.shape[1] == self.image_width
        return crop

    def _mean(self, tile_size):
        return np.ones((tile_size, tile_size), dtype=np.float32)

    def _pyramid(self, tile_size):
        w, _, _ = compute_patch_weight_loss(tile_size, tile_size)
        return w

    # ---- geometry for the device path -------------------------------------------------------------------------
    def grid(self):
        """(tiles per row, tiles per column) of the regular crop grid."""
        nx = len(set(c[0] for c in self.crops))
        return nx, len(self.crops) // nx
