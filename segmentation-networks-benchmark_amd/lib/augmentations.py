"""The test-time-augmentation pair of the reference's ``lib.augmentations`` (/root/reference/lib/augmentations.py:
476-511): the 8 elements of D4 per image and their inverse + average.  (The training-time augmenters of that module
are data plumbing: out of scope)  The device path (segnb.tiled) folds both into index maps built from the
same table."""
import numpy as np

# the dihedral group of the square in the order the reference emits it: (quarter turns, mirrored afterwards)
D4 = tuple((k, f) for f in (False, True) for k in range(4))


def d4_apply(image, k, mirrored):
    out = np.rot90(image, k)
    return np.fliplr(out) if mirrored else out


def d4_undo(image, k, mirrored):
    return np.rot90(np.fliplr(image) if mirrored else image, -k)


def tta_d4_aug(images):
    """[img, ...] -> [the 8 D4 views of img0, the 8 views of img1, ...] (augmentations.py:476-491)"""
    return [d4_apply(image, k, f) for image in images for k, f in D4]


def tta_d4_deaug(image_list):
    """Inverse of tta_d4_aug followed by the mean over each group of 8 (augmentations.py:494-511); the partial sums
    run in table order, as there."""
    assert len(image_list) % len(D4) == 0
    res = []
    for first in range(0, len(image_list), len(D4)):
        acc = None
        for view, (k, f) in zip(image_list[first:first + len(D4)], D4):
            back = d4_undo(view, k, f)
            acc = back if acc is None else acc + back
        res.append(acc * float(1. / 8.))
    return res
