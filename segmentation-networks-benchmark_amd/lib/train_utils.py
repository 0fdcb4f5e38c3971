"""Training helpers -- drop-in for the reference's ``lib.train_utils`` (/root/reference/lib/train_utils.py:14-125):
``AverageMeter``, ``find_optimal_lr``, ``auto_file``, ``PRCurveMeter`` with the same names, signatures and
results.  ``PRCurveMeter.update`` is one device histogram kernel instead of 127 host passes over the prediction map."""
import glob
import os

import numpy as np
import torch


class AverageMeter(object):
    """Running value / sum / count / average (train_utils.py:14-33)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return '%.3f' % self.avg


def find_optimal_lr(model, criterion, optimizer, dataloader):
    """LR range test (train_utils.py:36-69): 30 steps, lr = 1e-8 * 2^i, the step body of the training loop.
    Like the reference it never zeroes gradients between steps (they accumulate) and applies the schedule
    through LambdaLR (so the optimizer's base lr multiplies the table, as there).

    DEVIATION, deliberate: the reference calls ``scheduler.step()`` BEFORE every step
    (train_utils.py:49-55), so its step i runs at ``lrs[i + 1]`` while the returned table pairs ``loss[i]`` with
    ``lrs[i]``, and its last step indexes ``lrs[30]`` -- one past the 30-entry table (an IndexError on the torch versions
    this was checked against).  Here step i runs at ``lrs[i]``, the pairing the returned arrays state, and the
    schedule is clamped at the last entry."""
    from torch.optim.lr_scheduler import LambdaLR
    lrs = np.array([1e-8 * 2.0 ** i for i in range(30)], dtype=np.float32)
    loss = np.zeros_like(lrs)
    scheduler = LambdaLR(optimizer, lr_lambda=lambda i: float(lrs[min(i, len(lrs) - 1)]))
    device = next(model.parameters()).device
    with torch.set_grad_enabled(True):
        model.train()
        it = iter(dataloader)
        for i in range(len(lrs)):
            if i > 0:
                scheduler.step()
            x, y = next(it)
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            batch_loss = criterion(model(x), y)
            (x.size(0) * batch_loss).backward()
            optimizer.step()
            loss[i] = batch_loss.cpu().item()
    return lrs, loss


def auto_file(filename, where='.'):
    """Locate a uniquely named file below `where` (train_utils.py:72-89)."""
    direct = os.path.join(where, filename)
    if os.path.isfile(direct):
        return filename
    hits = list(glob.iglob(os.path.join(where, '**', filename), recursive=True))
    if not hits:
        raise FileNotFoundError('Given file could not be found with recursive search:' + filename)
    if len(hits) > 1:
        raise FileNotFoundError('More than one file matches given filename. Please specify it explicitly' + filename)
    return hits[0]


class PRCurveMeter(object):
    """Confusion counts at n_thresholds probability thresholds arange(0, 1, 1/n) (train_utils.py:92-131)."""

    def __init__(self, n_thresholds=127):
        self.n_thresholds = n_thresholds
        self.k = 2
        self.thresholds = np.arange(0., 1., 1. / n_thresholds, dtype=np.float32)
        self.tp = np.zeros(n_thresholds, dtype=np.uint64)
        self.tn = np.zeros(n_thresholds, dtype=np.uint64)
        self.fp = np.zeros(n_thresholds, dtype=np.uint64)
        self.fn = np.zeros(n_thresholds, dtype=np.uint64)

    def reset(self):
        for a in (self.tp, self.tn, self.fp, self.fn):
            a.fill(0)

    def update(self, y_pred, y_true):
        """One device histogram launch (segnb_pr_histogram): bucket b of a pixel = number of thresholds strictly below
        sigmoid(logit) = number of thresholds it is predicted positive at; the 127 confusion matrices of the
        reference's threshold loop (train_utils.py:113-125) are suffix sums of the two class histograms."""
        from segnb import _native as nv
        x = y_pred.detach().contiguous().float().reshape(-1)
        t = y_true.detach().reshape(-1)
        if t.dtype != torch.int64:
            t = t.to(torch.int64)
        t = t.to(x.device).contiguous()        # (the reference takes both through .cpu(): any device mix is accepted)
        n = self.n_thresholds
        if not x.is_cuda and not nv.has_test_backend():
            # host tensors (the reference's own call pattern, train_utils.py:111-112): the same histogram with torch ops --
            # the kernel dereferences device pointers
            thr = torch.from_numpy(self.thresholds)
            bucket = torch.bucketize(torch.sigmoid(x), thr, right=False)       # thresholds strictly below the probability
            hist = torch.stack([torch.bincount(bucket[t == 0], minlength=n + 1),
                                torch.bincount(bucket[t != 0], minlength=n + 1)])
            self._add_hist(hist.numpy().astype(np.uint64))
            return
        thr = torch.from_numpy(self.thresholds).to(x.device)
        hist = torch.zeros((2, n + 1), dtype=torch.int64, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
        nv.call('segnb_pr_histogram', nv.ptr(x), nv.ptr(t), x.numel(), nv.ptr(thr), n, nv.ptr(hist), st)
        self._add_hist(hist.cpu().numpy().astype(np.uint64))

    def _add_hist(self, h):
        hn, hp = h[0], h[1]
        # predicted positive at threshold i  <=>  bucket > i
        tp = hp[::-1].cumsum()[::-1][1:]
        fp = hn[::-1].cumsum()[::-1][1:]
        self.tp += tp
        self.fp += fp
        self.fn += hp.sum() - tp
        self.tn += hn.sum() - fp

    def precision(self):
        return np.divide(self.tp, self.tp + self.fp)

    def recall(self):
        return np.divide(self.tp, self.tp + self.fn)
