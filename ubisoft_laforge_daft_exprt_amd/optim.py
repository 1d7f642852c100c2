"""Fused optimiser step: Adam + global-norm clipping + the reference LR schedule.

Reference behaviour: ``Adam(betas=[0.9, 0.98], eps=1e-9, weight_decay=1e-6, amsgrad=False)`` (src/daft_exprt/train.py:278-280),
``clip_grad_norm_(model.parameters(), grad_clip_thresh)`` (:443), linear warm-up then inverse-sqrt decay (:148-160).
Parameters, gradients and both moments live in flat buckets (the gradient buckets are ddp.GradientReducer's communication
buffers), so one ``dx_adam_step`` launch per bucket replaces ~10 ATen launches per parameter tensor.
"""
from __future__ import annotations

import math

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
        self.pflat, self.m, self.v = [], [], []
        for bucket, gflat in zip(reducer.buckets, reducer.flat):
            pflat = torch.empty_like(gflat)
            off = 0
            for p in bucket:                       # re-home every parameter inside the flat bucket (same offsets as its gradient)
                n = p.numel()
                pflat[off:off + n].copy_(p.data.reshape(-1))
                p.data = pflat[off:off + n].view_as(p)
                off += n
            self.pflat.append(pflat)
            self.m.append(torch.zeros_like(gflat))
            self.v.append(torch.zeros_like(gflat))
        self.normsq = torch.zeros(1, dtype=torch.float32, device=reducer.flat[0].device)

    def step(self, lr=None):
        """Call after ``reducer.finish()``.  Returns the global gradient norm (device scalar, no host sync)."""
        if lr is not None:
            self.lr = lr
        self.step_count += 1
        self.normsq.zero_()
        for g in self.reducer.flat:
            lib().dx_sumsq(_p(g), g.numel(), _p(self.normsq), _stream())
        for p, g, m, v in zip(self.pflat, self.reducer.flat, self.m, self.v):
            lib().dx_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(self.lr), self.betas[0], self.betas[1], float(self.eps),
                               float(self.weight_decay), self.step_count, _p(self.normsq), float(self.max_norm), _stream())
        ops.invalidate_packs()                      # parameters were written behind autograd's back: force a re-pack ...
        ops.repack_all()                            # ... which is one launch for the whole model
        return self.normsq.sqrt()

    def state_dict(self):
        return {'step': self.step_count, 'lr': self.lr, 'm': [t.clone() for t in self.m], 'v': [t.clone() for t in self.v]}

    def load_state_dict(self, state):
        self.step_count, self.lr = state['step'], state['lr']
        for dst, src in zip(self.m, state['m']):
            dst.copy_(src)
        for dst, src in zip(self.v, state['v']):
            dst.copy_(src)
