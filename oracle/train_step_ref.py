"""CPU restatement of the training-step body, torch_train.py:176-190.  TEST INFRASTRUCTURE.

    optimizer.zero_grad()                    :180
    outputs = model(x)                       :183
    batch_loss = loss(outputs, y)            :185
    (batch_size * batch_loss).backward()     :187-188   <- note the xB gradient scale
    optimizer.step()                         :190       plain SGD (torch_train.py:71)

Runs on a state_dict with torch-CPU autograd; returns loss, logits and grads.
Also the ``cpu_baseline`` ("port") timed by bench.py.
"""
import torch

from . import losses_ref, zf_unet_ref


def loss_and_grads(sd, x, y, loss_name='bce_jaccard', drop=None, train=True, forward=zf_unet_ref.forward,
                   is_param=zf_unet_ref.is_param, autocast=False):
    """One forward + loss + (B*loss).backward().  Returns (loss, logits, {name: grad}).
    autocast: run the model under torch.autocast('cpu', bfloat16) -- what the reference's own code computes when a user
    switches it to bf16 (fp32 parameters, bf16 convolutions, loss on the fp32-cast logits): the yardstick the HIP bf16
    throughput path is measured against (tests/test_zf_unet_gpu.py)."""
    leaves = {}
    work = {}
    for k, v in sd.items():
        if is_param(k):
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
        else:
            work[k] = v            # BN buffers: updated in place, as nn.BatchNorm2d does
    if autocast:
        with torch.autocast('cpu', dtype=torch.bfloat16):
            logits = forward(work, x, train=train, drop=drop)
        logits = logits.float()
    else:
        logits = forward(work, x, train=train, drop=drop)
    loss = losses_ref.LOSSES[loss_name](logits, y)
    (x.shape[0] * loss).backward()
    grads = {k: p.grad for k, p in leaves.items()}
    return loss.detach(), logits.detach(), grads


def sgd_apply(sd, grads, lr):
    with torch.no_grad():
        for k, g in grads.items():
            sd[k].add_(g, alpha=-lr)


def train_step(sd, x, y, loss_name='bce_jaccard', lr=1e-3, drop=None, **kw):
    """Full step incl. plain-SGD update of ``sd`` in place.  Returns (loss, logits, grads)."""
    loss, logits, grads = loss_and_grads(sd, x, y, loss_name, drop, True, **kw)
    sgd_apply(sd, grads, lr)
    return loss, logits, grads


def synthetic_batch(batch, size, seed=1234, channels=3):
    """SURVEY 8d synthetic inputs: x = randn(B,3,S,S); y = (rand(B,1,S,S) > 0.7).long()."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, channels, size, size, generator=g)
    y = (torch.rand(batch, 1, size, size, generator=g) > 0.7).long()
    return x, y
