"""Optimizers for the hot path.  ``SGD`` / ``RMSprop`` / ``Adam`` are the torch.optim classes of the reference's
``get_optimizer`` (torch_train.py:67-79: plain SGD, RMSprop and Adam with torch defaults) whose ``step`` is ONE
launch over the model's flat parameter buffer when every parameter and gradient lives in segnb's FlatParams;
anything else (momentum, weight decay, amsgrad, foreign params, sparse grads) takes torch's own implementation."""
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


def _flat_of(group, simple):
    if not simple:
        return None
    flat, n = None, 0
    for p in group['params']:
        f = FlatParams.registry.get(id(p))
        if f is None or (flat is not None and f is not flat) or p.grad is None:
            return None
        flat = f
        n += 1
    if flat is None or n != len(flat._off) or not flat.grads_alias():
        return None
    return flat


def _stream(flat):
    return torch.cuda.current_stream(flat.flat_p.device).cuda_stream if flat.flat_p.is_cuda else 0


class _FlatStateMixin(object):
    """flat state buffers live on the optimizer (keyed by the FlatParams object); parameters torch would handle
    itself fall through to the parent class"""

    def _flat_state(self, flat, names):
        st = self.__dict__.setdefault('_segnb_state', {})
        key = id(flat)
        if key not in st or st[key]['p'].data_ptr() != flat.flat_p.data_ptr():
            st[key] = dict(p=flat.flat_p, step=0, **{n: torch.zeros_like(flat.flat_p) for n in names})
        return st[key]

    def _step_groups(self, closure, handler):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plain = [g for g in self.param_groups if not handler(g)]
        if plain:
            saved = self.param_groups
            self.param_groups = plain
            try:
                super(_FlatStateMixin, self).step()
            finally:
                self.param_groups = saved
        return loss


class RMSprop(_FlatStateMixin, torch.optim.RMSprop):
    @torch.no_grad()
    def step(self, closure=None):
        def handle(g):
            flat = _flat_of(g, g['momentum'] == 0 and g['weight_decay'] == 0 and not g['centered'] and
                            not g.get('maximize'))
            if flat is None:
                return False
            st = self._flat_state(flat, ('square_avg',))
            nv.call('segnb_rmsprop_step', nv.ptr(flat.flat_p), nv.ptr(flat.flat_g), nv.ptr(st['square_avg']), flat.total,
                    float(g['lr']), float(g['alpha']), float(g['eps']), _stream(flat))
            flat.version += 1
            return True
        return self._step_groups(closure, handle)


class Adam(_FlatStateMixin, torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        def handle(g):
            flat = _flat_of(g, g['weight_decay'] == 0 and not g['amsgrad'] and not g.get('maximize'))
            if flat is None:
                return False
            st = self._flat_state(flat, ('exp_avg', 'exp_avg_sq'))
            st['step'] += 1
            b1, b2 = g['betas']
            nv.call('segnb_adam_step', nv.ptr(flat.flat_p), nv.ptr(flat.flat_g), nv.ptr(st['exp_avg']),
                    nv.ptr(st['exp_avg_sq']), flat.total, float(g['lr']), float(b1), float(b2), float(g['eps']),
                    st['step'], _stream(flat))
            flat.version += 1
            return True
        return self._step_groups(closure, handle)
