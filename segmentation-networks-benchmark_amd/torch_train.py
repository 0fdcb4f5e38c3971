"""Harness for the hot path: the factories and the train / validate step bodies of the reference's driver
(/root/reference/torch_train.py:67-148, :159-215, :240-285), without its dataset / tensorboard / argparse plumbing
(out of scope).  Same factory names and string keys, same step order:

    optimizer.zero_grad(); outputs = model(x); batch_loss = loss(outputs, y)
    (batch_size * batch_loss).backward(); optimizer.step()                         torch_train.py:180-190
"""
import sys

import torch

from lib.losses import (BCEAndDiceLoss, BCEWithLogitsLossAndSmoothJaccard, BCEWithSigmoidLoss, FocalLossBinary,
                        JaccardLoss)
from lib.metrics import JaccardScore, PixelAccuracy
from lib.train_utils import AverageMeter
from segnb import optim as segnb_optim


def get_optimizer(optimizer_name, model_parameters, learning_rate):
    name = optimizer_name.lower()
    if name == 'sgd':
        return segnb_optim.SGD(model_parameters, lr=learning_rate)     # torch.optim.SGD with a one-launch step
    if name == 'rms':
        return segnb_optim.RMSprop(model_parameters, lr=learning_rate)
    if name == 'adam':
        return segnb_optim.Adam(model_parameters, lr=learning_rate)
    raise ValueError(optimizer_name)


def get_loss(loss):
    key = loss.lower()
    table = {'jaccard': JaccardLoss, 'bce_jaccard': BCEWithLogitsLossAndSmoothJaccard, 'bce': BCEWithSigmoidLoss,
             'bce_dice': BCEAndDiceLoss, 'focal': lambda: FocalLossBinary(size_average=False)}
    if key not in table:
        raise ValueError(loss)
    return table[key]()


def get_model(model_name, patch_size=None, num_channels=3):
    name = str.lower(model_name)
    if name == 'zf_unet':
        from lib.models.zf_unet import ZF_UNET
        return ZF_UNET()
    if name == 'unet16':
        from lib.models.unet16 import UNet16
        return UNet16(pretrained=True)          # "not 'vgg'" -> random init, exactly as the reference (:113)
    if name == 'linknet34':
        from lib.models.linknet import LinkNet34
        return LinkNet34(pretrained=True, num_channels=num_channels, num_classes=1)
    if name == 'tiramisu67':
        from lib.models.tiramisu import FCDenseNet67
        return FCDenseNet67(n_classes=1)
    if name == 'tiramisu103':                    # defined by the reference (tiramisu.py:201) but not wired there
        from lib.models.tiramisu import FCDenseNet103
        return FCDenseNet103(n_classes=1)
    raise ValueError(model_name)


def default_metrics():
    return {'iou': JaccardScore(), 'accuracy': PixelAccuracy()}


def grad_global_abs_max(model):
    """'train/grad/global_abs_max' of torch_train.py:199-205.  One fused reduction over the flat gradient buffer and
    one host sync when the model's gradients live in segnb's FlatParams; the reference's per-tensor loop otherwise."""
    from segnb.engine import FlatParams
    params = [p for p in model.parameters() if p.grad is not None]
    flat = FlatParams.registry.get(id(params[0])) if params else None
    if flat is not None and len(params) == len(flat._off) and flat.grads_alias():
        return flat.grad_absmax().cpu().item()
    grad_max = 0
    for p in params:
        grad_max = max(grad_max, p.grad.abs().max().cpu().item())
    return grad_max


def train(model, loss, optimizer, dataloader, epoch=0, metrics=None, grad_monitor=None):
    """One epoch of the training step; returns (AverageMeter of the loss, {metric: AverageMeter}).
    grad_monitor: optional callable(step, value) receiving the global gradient abs-max of every step (the
    summary_writer.add_scalar('train/grad/global_abs_max', ...) of torch_train.py:205)."""
    metrics = metrics or {}
    losses, scores = AverageMeter(), {k: AverageMeter() for k in metrics}
    device = next(model.parameters()).device
    with torch.set_grad_enabled(True):
        model.train()
        for x, y in dataloader:
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            optimizer.zero_grad()
            outputs = model(x)
            batch_loss = loss(outputs, y)
            (x.size(0) * batch_loss).backward()
            optimizer.step()
            losses.update(batch_loss.cpu().item())
            if grad_monitor is not None:
                grad_monitor(losses.count - 1, grad_global_abs_max(model))
            for k, m in metrics.items():
                scores[k].update(m(outputs, y).cpu().item())      # answered from the loss launch's sums
    return losses, scores


def validate(model, loss, dataloader, epoch=0, metrics=None):
    metrics = metrics or {}
    losses, scores = AverageMeter(), {k: AverageMeter() for k in metrics}
    device = next(model.parameters()).device
    with torch.set_grad_enabled(False):
        model.eval()
        for x, y in dataloader:
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            outputs = model(x)
            losses.update(loss(outputs, y).cpu().item())
            for k, m in metrics.items():
                scores[k].update(m(outputs, y).cpu().item())
    return losses, scores


def save_snapshot(model, optimizer, loss, epoch, train_history, snapshot_file):
    """torch_train.py:308-316: the reference's checkpoint dict, key for key.  train_history: a pandas DataFrame (as
    there) or anything with ``to_dict``; a plain dict is stored as is."""
    torch.save({
        'model': model.state_dict(),
        'optimizer': optimizer.state_dict(),
        'epoch': epoch,
        'loss': loss,
        'train_history': train_history.to_dict() if hasattr(train_history, 'to_dict') else train_history,
        'args': ' '.join(sys.argv[1:]),
    }, snapshot_file)


def restore_snapshot(model, optimizer, snapshot_file):
    """torch_train.py:319-330 -> (start_epoch, train_history DataFrame -- the stored dict where pandas is absent --, best_loss).  Checkpoints written by the
    reference load here and vice versa (same keys, same state_dict names)."""
    checkpoint = torch.load(snapshot_file, map_location='cpu', weights_only=False)
    start_epoch = checkpoint['epoch'] + 1
    best_loss = checkpoint['loss']
    model.load_state_dict(checkpoint['model'])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint['optimizer'])
    try:                                    # (the reference returns a DataFrame; without pandas the stored dict is returned as is)
        import pandas as pd
        train_history = pd.DataFrame.from_dict(checkpoint['train_history'])
    except ImportError:
        train_history = checkpoint['train_history']
    return start_epoch, train_history, best_loss
