"""Optimizers for the hot path.  ``SGD`` / ``RMSprop`` / ``Adam`` are the torch.optim classes of the reference's
``get_optimizer`` (torch_train.py:67-79: plain SGD, RMSprop and Adam with torch defaults) whose ``step`` is ONE
launch over the model's flat parameter buffer when every parameter and gradient lives in segnb's FlatParams;
anything else (momentum, weight decay, amsgrad, foreign params, sparse grads) takes torch's own implementation."""
import os

import torch

from . import _native as nv
from .engine import FlatParams


def _take_fused_token(flat):
    """True when segnb.dist.DataParallel already applied this step's update bucket by bucket behind the all-reduces
    (DataParallel.fuse_optimizer): step() then only has to leave the parameters' version bumped."""
    if getattr(flat, 'stepped_in_backward', False):
        flat.stepped_in_backward = False
        return True
    return False


class SGD(torch.optim.SGD):
    # fuse_pack = False (class attribute / SEGNB_SGD_PACK=0): the update as one launch over the flat buffers, the pack at the next forward (A/B)
    fuse_pack = os.environ.get('SEGNB_SGD_PACK', '1') != '0'

    # ---- update of one contiguous range of the flat buffers (DataParallel.fuse_optimizer) -------------------------------
    def fusable_group(self, flat):
        """The param group this optimizer would update with its one-launch step over `flat`, else None (grad aliasing is
        checked by the caller: during backward .grad may still be None)."""
        return _sole_group(self, flat, lambda g: not (g['momentum'] != 0 or g['weight_decay'] != 0 or g['nesterov'] or
                                                      g.get('maximize')))

    @torch.no_grad()
    def step_range(self, flat, group, start, end, first):
        st = torch.cuda.current_stream(flat.flat_p.device).cuda_stream if flat.flat_p.is_cuda else 0
        nv.call('segnb_sgd_step', nv.ptr(flat.flat_p[start:end]), nv.ptr(flat.flat_g[start:end]), end - start,
                float(group['lr']), st)

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
            if _take_fused_token(flat):
                flat.version += 1
                continue
            # the model plan may apply the update itself, fused with the weight pack of the next forward (one read of every
            # convolution weight instead of two: segnb_sgd_pack_pair_multi); same arithmetic, bit-identical parameters
            hook = flat.sgd_pack_hook if self.fuse_pack else None
            if hook is not None and hook(float(group['lr'])):
                flat.version += 1
                flat.sgd_pack_done()
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
    """The moments live in flat buffers parallel to FlatParams.flat_p (one launch per step), and are PUBLISHED through
    ``optimizer.state`` as per-parameter views of those buffers plus torch's ``step`` entry, so ``state_dict()`` /
    ``load_state_dict()`` round-trip exactly like torch.optim's (the reference's save_snapshot / restore_snapshot,
    torch_train.py:308-330, checkpoint ``optimizer.state_dict()``).  State loaded from a checkpoint -- or produced by
    the parent class before the flat path became eligible -- is adopted: copied into the flat buffers on the next step.
    Parameters torch has to handle itself fall through to the parent class."""

    def _flat_state(self, flat, names):
        st = self.__dict__.setdefault('_segnb_state', {})
        key = id(flat)
        ent = st.get(key)
        if ent is None or ent['p'].data_ptr() != flat.flat_p.data_ptr():
            ent = dict(p=flat.flat_p, step=torch.zeros((), dtype=torch.float32),
                       **{n: torch.zeros_like(flat.flat_p) for n in names})
            st[key] = ent
        params = flat.param_list()
        first = self.state.get(params[0])
        n0 = names[0]
        synced = (first is not None and first.get('step') is ent['step'] and n0 in first and
                  first[n0].data_ptr() == ent[n0].data_ptr() + 4 * flat._off[id(params[0])][0])
        if not synced:
            for p in params:
                ps = self.state[p]
                off, n = flat._off[id(p)]
                for name in names:
                    view = ent[name][off:off + n].view(p.shape)
                    cur = ps.get(name)
                    if cur is not None and cur.data_ptr() != view.data_ptr():
                        view.copy_(cur)                      # adopt loaded / foreign state
                    ps[name] = view
                cur = ps.get('step')
                if cur is not None and cur is not ent['step']:
                    ent['step'].fill_(float(cur))
                ps['step'] = ent['step']                     # ONE shared counter tensor (state_dict copies it per param)
        return ent

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


def _sole_group(opt, flat, simple):
    if len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if not simple(g) or len(g['params']) != len(flat._off):
        return None
    if any(FlatParams.registry.get(id(p)) is not flat for p in g['params']):
        return None
    return g


class RMSprop(_FlatStateMixin, torch.optim.RMSprop):
    @staticmethod
    def _simple(g):
        return g['momentum'] == 0 and g['weight_decay'] == 0 and not g['centered'] and not g.get('maximize')

    def fusable_group(self, flat):
        return _sole_group(self, flat, self._simple)

    @torch.no_grad()
    def step_range(self, flat, g, start, end, first):
        st = self._flat_state(flat, ('square_avg',))
        if first:
            st['step'] += 1
        nv.call('segnb_rmsprop_step', nv.ptr(flat.flat_p[start:end]), nv.ptr(flat.flat_g[start:end]),
                nv.ptr(st['square_avg'][start:end]), end - start, float(g['lr']), float(g['alpha']), float(g['eps']),
                _stream(flat))

    @torch.no_grad()
    def step(self, closure=None):
        def handle(g):
            flat = _flat_of(g, self._simple(g))
            if flat is None:
                return False
            if _take_fused_token(flat):
                flat.version += 1
                return True
            st = self._flat_state(flat, ('square_avg',))
            st['step'] += 1
            nv.call('segnb_rmsprop_step', nv.ptr(flat.flat_p), nv.ptr(flat.flat_g), nv.ptr(st['square_avg']), flat.total,
                    float(g['lr']), float(g['alpha']), float(g['eps']), _stream(flat))
            flat.version += 1
            return True
        return self._step_groups(closure, handle)


class Adam(_FlatStateMixin, torch.optim.Adam):
    @staticmethod
    def _simple(g):
        return g['weight_decay'] == 0 and not g['amsgrad'] and not g.get('maximize')

    def fusable_group(self, flat):
        return _sole_group(self, flat, self._simple)

    @torch.no_grad()
    def step_range(self, flat, g, start, end, first):
        st = self._flat_state(flat, ('exp_avg', 'exp_avg_sq'))
        if first:
            st['step'] += 1
        b1, b2 = g['betas']
        nv.call('segnb_adam_step', nv.ptr(flat.flat_p[start:end]), nv.ptr(flat.flat_g[start:end]),
                nv.ptr(st['exp_avg'][start:end]), nv.ptr(st['exp_avg_sq'][start:end]), end - start, float(g['lr']),
                float(b1), float(b2), float(g['eps']), int(st['step']), _stream(flat))

    @torch.no_grad()
    def step(self, closure=None):
        def handle(g):
            flat = _flat_of(g, self._simple(g))
            if flat is None:
                return False
            if _take_fused_token(flat):
                flat.version += 1
                return True
            st = self._flat_state(flat, ('exp_avg', 'exp_avg_sq'))
            st['step'] += 1
            b1, b2 = g['betas']
            nv.call('segnb_adam_step', nv.ptr(flat.flat_p), nv.ptr(flat.flat_g), nv.ptr(st['exp_avg']),
                    nv.ptr(st['exp_avg_sq']), flat.total, float(g['lr']), float(b1), float(b2), float(g['eps']),
                    int(st['step']), _stream(flat))
            flat.version += 1
            return True
        return self._step_groups(closure, handle)
