"""Fused Adam on the networks' flat parameter buffers.

Same numerics as ``torch.optim.Adam`` in the configuration the reference uses (train.py:148-149,195:
betas (0.0, 0.99), eps 1e-8, no weight decay, no amsgrad; torch-2.10 update form) with per-parameter
step counters, so that a parameter without a gradient is skipped entirely (no moment decay, no
bias-correction step) — which is what happens to not-yet-grown blocks in the reference.
Contiguous runs of active parameters that share a step count are updated by ONE ``pg_adam`` launch.
It is a ``torch.optim.Optimizer`` so ``LambdaLR`` and the ``LRScheduler`` plugin work unchanged."""
import math

import torch

from . import ops


def _padded(n):
    return (n + 3) // 4 * 4


def _flat_view(t, n):
    """1-D view of ``n`` floats starting at t's first element (t is a view into a flat buffer)."""
    return torch.as_strided(t, (n,), (1,))


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 grad_scale=1.0):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('weight_decay / amsgrad are not used by the reference configuration')
        super(FusedAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale          # 1/world_size when gradients were SUM-all-reduced
        self._nets = []                       # networks whose derived (backward-data) weight copies go stale
        self._flat = {}                       # id(group) -> (base_ptr, m_flat, v_flat)
        self._probe = {}                      # id(group) -> (address of the group's first parameter, number of parameters) at the last check

    def attach(self, *nets):
        self._nets.extend(nets)
        return self

    def _flat_state(self, gi, group):
        params = group['params']
        probe = params[0].data_ptr()              # a re-flatten moves every parameter: the first one's address tells
        hit = self._probe.get(gi)
        if hit is not None and hit[0] == probe and len(params) == hit[1] and gi in self._flat:
            return self._flat[gi]
        base = min(p.data_ptr() for p in params)
        self._probe[gi] = (probe, len(params))
        if gi in self._flat:
            if self._flat[gi][0] == base:
                return self._flat[gi]
            return self._rebase(gi, group, base)          # the network was re-flattened (.to() / .cuda() / .float())
        return self._new_flat_state(gi, group, base)

    def _rebase(self, gi, group, base):
        """The parameters moved to a new flat buffer: rebuild the flat moments at the new location and carry the old
        ones over (per-parameter offsets inside the buffer are unchanged by a re-flatten)."""
        old_base, m_old, v_old = self._flat.pop(gi)
        entry = self._new_flat_state(gi, group, base)
        _, m_new, v_new = entry
        for p in group['params']:
            st = self.state.get(p)
            if not st:
                continue
            off = (p.data_ptr() - base) // 4
            for key, flat in (('exp_avg', m_new), ('exp_avg_sq', v_new)):
                view = flat[off:off + p.numel()].view(p.shape)
                view.copy_(st[key].to(view.device))
                st[key] = view
        return entry

    def _new_flat_state(self, gi, group, base):
        params = group['params']
        top = max(p.data_ptr() + _padded(p.numel()) * 4 for p in params)
        total = (top - base) // 4
        if total != sum(_padded(p.numel()) for p in params):
            raise RuntimeError('FusedAdam expects the parameters of one network (views of its flat buffer)')
        dev = params[0].device
        entry = (base, torch.zeros(total, dtype=torch.float32, device=dev),
                 torch.zeros(total, dtype=torch.float32, device=dev))
        from .network import network_of_flat_ptr
        net = network_of_flat_ptr(base)
        if net is not None and net not in self._nets:
            self._nets.append(net)
        self._flat[gi] = entry
        return entry

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Resume: the loaded moments are copied into the flat m / v buffers (views are re-created) so that
        contiguous runs keep updating with one launch.  (The reference never saved optimizer state,
        plugins.py:142-174; this is the 'next row' checkpoint extension of SURVEY.md §8f.)"""
        super(FusedAdam, self).load_state_dict(state_dict)
        self._flat = {}
        self._probe = {}
        for gi, group in enumerate(self.param_groups):
            base, mflat, vflat = self._flat_state(gi, group)
            for p in group['params']:
                st = self.state.get(p)
                if not st:
                    continue
                off = (p.data_ptr() - base) // 4
                for key, flat in (('exp_avg', mflat), ('exp_avg_sq', vflat)):
                    v = flat[off:off + p.numel()].view(p.shape)
                    v.copy_(st[key].to(v.device))
                    st[key] = v
                st['step'] = int(st['step'])

    @torch.no_grad()
    def step(self, closure=None):
        from . import engine
        for net in self._nets:                # (a side-stream refresh of derived weights may still be READING the parameters)
            engine._await_backward_copies(net)
        for gi, group in enumerate(self.param_groups):
            lr, (b1, b2), eps = float(group['lr']), group['betas'], group['eps']
            base, mflat, vflat = self._flat_state(gi, group)
            runs = []
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    off = (p.data_ptr() - base) // 4
                    st['step'] = 0
                    st['exp_avg'] = mflat[off:off + p.numel()].view(p.shape)
                    st['exp_avg_sq'] = vflat[off:off + p.numel()].view(p.shape)
                st['step'] += 1
                runs.append(p)
            runs.sort(key=lambda q: q.data_ptr())
            i = 0
            while i < len(runs):
                p0 = runs[i]
                t = self.state[p0]['step']
                start = p0.data_ptr()
                goff = p0.grad.data_ptr() - start
                end = start + _padded(p0.numel()) * 4
                j = i + 1
                while (j < len(runs) and runs[j].data_ptr() == end and self.state[runs[j]]['step'] == t
                       and runs[j].grad.data_ptr() - runs[j].data_ptr() == goff):
                    end += _padded(runs[j].numel()) * 4
                    j += 1
                last = runs[j - 1]
                n = (last.data_ptr() - start) // 4 + last.numel()
                st0 = self.state[p0]
                ops.adam(_flat_view(p0, n), _flat_view(p0.grad, n), _flat_view(st0['exp_avg'], n),
                         _flat_view(st0['exp_avg_sq'], n), lr, b1, b2, eps,
                         1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), self.grad_scale)
                i = j
        for net in self._nets:
            net.mark_params_changed()
        return None
