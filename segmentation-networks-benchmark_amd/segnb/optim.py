"""Optimizers for the hot path.  ``SGD`` is torch.optim.SGD (the reference's ``get_optimizer('sgd')``,
torch_train.py:70-71: plain SGD, no momentum / weight decay) whose ``step`` is ONE segnb_sgd_step launch
over the model's flat parameter buffer when every parameter and gradient lives in segnb's FlatParams;
anything else (momentum, foreign params, sparse grads) takes torch's own implementation."""
import torch

from . import _native as nv
from .engine import FlatParams


class SGD(torch.optim.SGD):
    def _flat_of_group(self, group):
        if group['momentum'] != 0 or group['weight_decay'] != 0 or group['nesterov'] or group.get('maximize'):
            return None
        flat = None
        n = 0
        for p in group['params']:
            f = FlatParams.registry.get(id(p))
            if f is None or (flat is not None and f is not flat) or p.grad is None:
                return None
            flat = f
            n += 1
        if flat is None or n != len(flat._off) or not flat.grads_alias():
            return None
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plain = []
        for group in self.param_groups:
            flat = self._flat_of_group(group)
            if flat is None:
                plain.append(group)
                continue
            st = torch.cuda.current_stream(flat.flat_p.device).cuda_stream if flat.flat_p.is_cuda else 0
            nv.call('segnb_sgd_step', nv.ptr(flat.flat_p), nv.ptr(flat.flat_g), flat.total, float(group['lr']), st)
            flat.version += 1
        if plain:
            saved = self.param_groups
            self.param_groups = plain
            try:
                super(SGD, self).step()
            finally:
                self.param_groups = saved
        return loss
