"""Fused optimiser step: Adam + global-norm clipping + the reference LR schedule.

Reference behaviour: ``Adam(betas=[0.9, 0.98], eps=1e-9, weight_decay=1e-6, amsgrad=False)`` (src/daft_exprt/train.py:278-280),
``clip_grad_norm_(model.parameters(), grad_clip_thresh)`` (:443), linear warm-up then inverse-sqrt decay (:148-160).
Parameters, gradients and both moments live in flat buckets (the gradient buckets are ddp.GradientReducer's communication
buffers), so one ``dx_adam_step`` launch per bucket replaces ~10 ATen launches per parameter tensor.

``state_dict()`` / ``load_state_dict()`` speak the ``torch.optim.Adam`` layout that the reference stores under the checkpoint's
``'optimizer'`` key (train.py:80-85, :134-138): ``{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params':
[0..n-1]}]}`` with ``i`` indexing ``filter(requires_grad, model.parameters())``, so optimiser resume is drop-in in both directions.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import lib
from .ops import _p, _stream


def update_learning_rate(hparams, iteration):
    """train.py:148-160"""
    lo, hi, warm = hparams.initial_learning_rate, hparams.max_learning_rate, hparams.warmup_steps
    if iteration < warm:
        return (hi - lo) / warm * iteration + lo
    return iteration ** -0.5 * hi / warm ** -0.5


class FusedAdam:
    def __init__(self, reducer, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, grad_clip_thresh=float('inf')):
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, tuple(betas), eps, weight_decay, grad_clip_thresh
        self.step_count = 0
        self.runtime = getattr(reducer.module, 'runtime', None) or ops.DEFAULT
        self.pflat, self.m, self.v = [], [], []
        self._slot = {}                                # parameter -> (bucket index, offset)
        # parameters and both moments mirror the reducer's single gradient allocation (same offsets): one launch steps everything
        self.p_all = torch.zeros_like(reducer.flat_all)
        self.m_all = torch.zeros_like(reducer.flat_all)
        self.v_all = torch.zeros_like(reducer.flat_all)
        for bi, (bucket, gflat) in enumerate(zip(reducer.buckets, reducer.flat)):
            o0 = reducer.bucket_offset[bi]
            pflat = self.p_all[o0:o0 + gflat.numel()]
            off = 0
            for p in bucket:                       # re-home every parameter inside the flat bucket (same offsets as its gradient)
                n = p.numel()
                pflat[off:off + n].copy_(p.data.reshape(-1))
                p.data = pflat[off:off + n].view_as(p)
                self._slot[p] = (bi, off)
                off += n
            self.pflat.append(pflat)
            self.m.append(self.m_all[o0:o0 + gflat.numel()])
            self.v.append(self.v_all[o0:o0 + gflat.numel()])
        # two squared-norm accumulators and two norm results, used alternately: the Adam launch of step k zeroes the accumulator of step
        # k + 1 (no fill launch per step) and the norm tensor handed back by step k stays valid until step k + 2 overwrites it
        self.normsq = torch.zeros(2, dtype=torch.float32, device=reducer.flat[0].device)
        self.norm = torch.zeros(2, dtype=torch.float32, device=reducer.flat[0].device)
        self.skipped = torch.zeros(1, dtype=torch.int32, device=reducer.flat[0].device)   # updates skipped for a non-finite gradient norm
        # torch.optim order: the trainable parameters in registration order (the reducer holds them reversed, in buckets)
        self.params = [p for p in reducer.module.parameters() if p.requires_grad]

    @property
    def param_groups(self):
        """Read-only view in torch.optim's shape (the reference reads ``param_groups[..]['lr']``, train.py:451-455)."""
        return [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay, 'amsgrad': False,
                 'params': list(range(len(self.params)))}]

    def step(self, lr=None, grad_scale=1.0):
        """Call after ``reducer.finish()``.  Returns the global gradient norm (device scalar, no host sync; the tensor is overwritten
        by the step after next).
        ``grad_scale``: the gradients in the buckets are multiplied by it on the fly (1 / loss scale in fp16 mode)."""
        if lr is not None:
            self.lr = lr
        self.step_count += 1
        k = self.step_count & 1
        nsq, other, norm = self.normsq[k:k + 1], self.normsq[1 - k:2 - k], self.norm[k:k + 1]
        g = self.reducer.flat_all                   # all buckets (the alignment gaps hold zeros: they add nothing and stay zero)
        lib().dx_sumsq(_p(g), g.numel(), _p(nsq), _stream())
        lib().dx_adam_step(_p(self.p_all), _p(g), _p(self.m_all), _p(self.v_all), g.numel(), float(self.lr), self.betas[0], self.betas[1], float(self.eps),
                           float(self.weight_decay), self.step_count, _p(nsq), float(self.max_norm), float(grad_scale), _p(self.skipped),
                           _p(norm), _p(other), _stream())
        self.runtime.invalidate_packs()             # parameters were written behind autograd's back: force a re-pack ...
        ops.repack_all(self.runtime)                # ... which is one launch for the whole model
        return norm[0]

    def skipped_steps(self) -> int:
        """Updates skipped because the (all-reduced) gradient norm was not finite: an overflow of the fp16 mode's loss scaling.  One host
        sync.  ``step_count`` (the bias-correction step) still counts a skipped update; after the first few hundred steps the bias
        corrections are 1 to within 1e-3 and the difference is immaterial."""
        return int(self.skipped.item())

    # -- checkpoint layout of torch.optim.Adam -------------------------------------------------------------------------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, p in enumerate(self.params):
                bi, off = self._slot[p]
                n = p.numel()
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': self.m[bi][off:off + n].view_as(p).clone(),
                            'exp_avg_sq': self.v[bi][off:off + n].view_as(p).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.params):
            raise ValueError(f'optimizer state has {sum(len(g["params"]) for g in groups)} parameters in {len(groups)} groups; '
                             f'this model has {len(self.params)} in one group')
        g = groups[0]
        self.lr = g.get('lr', self.lr)
        self.betas, self.eps, self.weight_decay = tuple(g.get('betas', self.betas)), g.get('eps', self.eps), g.get('weight_decay', self.weight_decay)
        steps = set()
        for m, v in zip(self.m, self.v):
            m.zero_()
            v.zero_()
        for idx, i in enumerate(g['params']):
            st = sd['state'].get(i)
            if st is None:
                continue
            p = self.params[idx]
            bi, off = self._slot[p]
            n = p.numel()
            self.m[bi][off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.v[bi][off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(st['step']))                       # torch >= 1.12 stores a tensor, 1.9 (the reference pin) an int
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps ONE bias-correction step')
        self.step_count = steps.pop() if steps else 0
        self.normsq.zero_()                                  # the step parity picks the accumulator: both start clean
