"""The test-time-augmentation pair of the reference's ``lib.augmentations`` (/root/reference/lib/augmentations.py:
476-511): the 8 elements of D4 per image and their inverse + average.  (The training-time augmenters of that module
are data plumbing: out of scope)  The device path (segnb.tiled) folds both into index maps."""
import numpy as np


def tta_d4_aug(images):
    res = []
    for image in images:
        res.extend([image, np.rot90(image, 1), np.rot90(image, 2), np.rot90(image, 3), np.fliplr(image),
                    np.fliplr(np.rot90(image, 1)), np.fliplr(np.rot90(image, 2)), np.fliplr(np.rot90(image, 3))])
    return res


def tta_d4_deaug(image_list):
    assert len(image_list) % 8 == 0
    res = []
    one_over_8 = float(1. / 8.)
    for i in range(0, len(image_list), 8):
        g = image_list[i:i + 8]
        res.append((g[0] + np.rot90(g[1], -1) + np.rot90(g[2], -2) + np.rot90(g[3], -3) + np.fliplr(g[4]) +
                    np.rot90(np.fliplr(g[5]), -1) + np.rot90(np.fliplr(g[6]), -2) +
                    np.rot90(np.fliplr(g[7]), -3)) * one_over_8)
    return res
