"""Deterministic state_dict fill shared by the fixture generator and the tests.  TEST INFRASTRUCTURE.

UNet16 / LinkNet34 carry 32 M / 22 M parameters: a fixture cannot ship them.  Instead both sides overwrite every
state_dict entry, in key order, with values drawn from a CPU generator seeded by (seed, position): the generator
script loads the fill into the REFERENCE module before running it, the tests load the same fill into the oracle
restatement and into the product module.  Scales are those of a healthy mid-training net (He-scaled kernels,
BatchNorm gamma around 1, non-trivial running statistics) so that no path is exercised at a degenerate value.
"""
import math
from collections import OrderedDict

import torch


def _tensor(name, shape, gen):
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == 'running_mean':
        return 0.1 * torch.randn(shape, generator=gen)
    if leaf == 'running_var':
        return 0.5 + torch.rand(shape, generator=gen)
    if len(shape) == 4:
        fan = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan)
    if leaf == 'weight':                               # BatchNorm / InPlaceABN gamma
        return 0.5 + torch.rand(shape, generator=gen)
    return 0.1 * torch.randn(shape, generator=gen)     # biases, beta


def seeded_state(shapes, seed):
    """shapes: ordered {key: shape} (a state_dict works too) -> OrderedDict of fresh tensors."""
    out = OrderedDict()
    for i, (k, v) in enumerate(shapes.items()):
        shape = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        gen = torch.Generator().manual_seed(seed * 1000003 + i)
        out[k] = _tensor(k, shape, gen)
    return out
