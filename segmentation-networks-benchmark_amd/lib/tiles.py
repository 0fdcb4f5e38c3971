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


def _axis_layout(length, tile_size, tile_step, image_margin):
    """One image axis -> (margin before, margin after, tile origins in the padded axis).

    Tiles of `tile_size` every `tile_step` must cover the padded axis exactly.  With image_margin == 0 the padding is
    whatever makes that true (the smaller half in front, tiles.py:66-77); a given image_margin must already make it
    true (tiles.py:79-89: ValueError otherwise)."""
    overlap = tile_size - tile_step
    if image_margin:
        if (length - overlap + 2 * image_margin) % tile_step:
            raise ValueError()
        before = after = image_margin
    else:
        count = max(1, math.ceil((length - overlap) / tile_step))
        extra = count * tile_step - (length - overlap)
        before = extra // 2
        after = extra - before
    origins = list(range(0, before + length + after - tile_size + 1, tile_step))
    return before, after, origins


class ImageSlicer:
    """Slices an image into overlapping square tiles and fuses per-tile predictions back (tiles.py:30-168): same
    constructor, attributes (``crops`` as (x, y, w, h), ``margin_left/right/top/bottom``, ``tile_size``,
    ``tile_step``, ``image_height/width``, ``compute_weight``) and results as the reference class."""

    def __init__(self, image_shape, tile_size, tile_step=0, image_margin=0, weight='mean'):
        self.image_height, self.image_width = image_shape[0], image_shape[1]
        self.tile_size, self.tile_step = tile_size, tile_step
        self.compute_weight = {'mean': self._mean, 'pyramid': self._pyramid}[weight]       # KeyError as the reference
        if not 1 <= tile_step <= tile_size:
            raise ValueError()
        self.margin_left, self.margin_right, xs = _axis_layout(self.image_width, tile_size, tile_step, image_margin)
        self.margin_top, self.margin_bottom, ys = _axis_layout(self.image_height, tile_size, tile_step, image_margin)
        self._grid = (len(xs), len(ys))
        self.crops = [(x, y, tile_size, tile_size) for y in ys for x in xs]                 # row-major, like :93-97
        self._coverage = None

    # ---- host (numpy) API of the reference ---------------------------------------------------------------
    def _canvas_shape(self, channels):
        return (self.margin_top + self.image_height + self.margin_bottom,
                self.margin_left + self.image_width + self.margin_right, channels)

    def _padded(self, image, borderType):
        if image.shape[:2] != (self.image_height, self.image_width):
            raise AssertionError('image is %s, slicer was built for %dx%d' % (image.shape[:2], self.image_height,
                                                                           self.image_width))
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

    def _overlap_add(self, canvas, pieces):
        """canvas[tile window k] += pieces[k], k in crop order (float64: the order fixes the rounding)."""
        for (x, y, tw, th), piece in zip(self.crops, pieces):
            canvas[y:y + th, x:x + tw] += piece
        return canvas

    def coverage(self):
        """Sum of the fusion weights of every tile covering a padded-image pixel, [Hp, Wp] float64, floored at the
        float64 epsilon (the divisor of merge, tiles.py:151-157); depends on the geometry only, so computed once."""
        if self._coverage is None:
            w = np.asarray(self.compute_weight(self.tile_size))[..., None]
            cov = self._overlap_add(np.zeros(self._canvas_shape(1), dtype=np.float64), [w] * len(self.crops))
            self._coverage = np.maximum(cov, np.finfo(np.float64).eps)
        return self._coverage

    def merge(self, tiles, dtype=np.float32):
        if len(tiles) != len(self.crops):
            raise ValueError
        channels = 1 if tiles[0].ndim == 2 else tiles[0].shape[2]
        w = np.asarray(self.compute_weight(self.tile_size))[..., None]
        ts = self.tile_size
        fused = self._overlap_add(np.zeros(self._canvas_shape(channels), dtype=np.float64),
                                  (np.reshape(t, (ts, ts, channels)) * w for t in tiles))
        fused = (fused / self.coverage()).astype(dtype)
        return fused[self.margin_top:self.margin_top + self.image_height,
                     self.margin_left:self.margin_left + self.image_width]

    def _mean(self, tile_size):
        return np.ones((tile_size, tile_size), dtype=np.float32)

    def _pyramid(self, tile_size):
        return compute_patch_weight_loss(tile_size, tile_size)[0]

    # ---- geometry for the device path -------------------------------------------------------------------------
    def grid(self):
        """(tiles per row, tiles per column) of the regular crop grid."""
        return self._grid
