"""The reference's optimiser-step structure around the HIP path ("boundary only" -- the trainer file itself
never travels, but a drop-in must reproduce what it does between two parameter updates).

One ``Trainer.train_step`` = src/daft_exprt/train.py:390-539 without logging / validation / checkpoint cadence:

    for each of ``accumulation_steps`` micro-batches:                       train.py:390-436
        parse_batch -> keep RAW frame energy / pitch for the consistency losses -> on-device batch conditioning
        -> model forward -> DaftExprtLoss(outputs, targets, iteration) -> (loss / accumulation_steps).backward()
    gradient exchange (bucketed RCCL all-reduce, launched during the LAST micro-batch's backward)        train.py:272 (DDP)
    clip_grad_norm_(parameters, grad_clip_thresh) + Adam(lr(iteration)) as ONE fused launch per bucket  train.py:443-445
    iteration += 1; lr = update_learning_rate(hparams, iteration)                                        train.py:517-539

Like the reference, a NaN loss does not stop the update (train.py:445-450 only skips LOGGING); ``nan_steps`` counts them.
The iteration counter starts at 1 (train.py:286) and the learning rate of step ``i`` is ``update_learning_rate(hparams, i)``.
``save_checkpoint`` / ``load_checkpoint`` use the reference's dict layout (train.py:63-145) including the torch.optim.Adam
optimiser layout.
"""
from __future__ import annotations

import math

import torch

from .ddp import GradientReducer
from .optim import FusedAdam, update_learning_rate


class Trainer:
    def __init__(self, model, criterion, hparams, conditioner=None, process_group=None, bucket_mb=16.0, grad_sink=True):
        self.model, self.criterion, self.hparams, self.conditioner = model, criterion, hparams, conditioner
        self.device = next(model.parameters()).device
        self.reducer = GradientReducer(model, bucket_mb=bucket_mb, process_group=process_group, grad_sink=grad_sink)
        self.optimizer = FusedAdam(self.reducer, lr=hparams.initial_learning_rate, betas=hparams.betas, eps=hparams.epsilon,
                                   weight_decay=hparams.weight_decay, grad_clip_thresh=hparams.grad_clip_thresh)
        self.accumulation_steps = int(getattr(hparams, 'accumulation_steps', 1))
        self.iteration = 1
        self.learning_rate = update_learning_rate(hparams, self.iteration)
        self.nan_steps = 0
        self.best_val_loss = float('inf')
        model.train()

    def _micro_batch(self, batch):
        """train.py:395-422: parse, condition, forward, loss."""
        inputs, targets = self.model.parse_batch(self.device, batch)
        raw_frames_energy, raw_frames_pitch = inputs[6], inputs[7]       # the frozen pitch predictor outputs RAW pitch
        if self.conditioner is not None:
            inputs = self.conditioner.process_batch(inputs, self.device)   # length tensors pass through (host lengths ride along)
        targets = (targets[0], inputs[3], inputs[4], targets[3], targets[4], targets[5], raw_frames_energy, raw_frames_pitch)
        outputs = self.model(inputs)
        return self.criterion(outputs, targets, self.iteration)

    def train_step(self, batches):
        """``batches``: the ``accumulation_steps`` micro-batches (reference 14-tuples) of one parameter update.
        Returns (summed loss tensor / accumulation_steps, list of per-micro-batch LossTerms, gradient norm tensor)."""
        batches = list(batches)
        if len(batches) != self.accumulation_steps:
            raise ValueError(f'expected {self.accumulation_steps} micro-batches, got {len(batches)}')
        self.reducer.zero_grad()
        tot, terms = None, []
        for k, batch in enumerate(batches):
            loss, indiv = self._micro_batch(batch)
            loss = loss / self.accumulation_steps
            with self.reducer.accumulate(sync=(k == len(batches) - 1)):
                loss.backward()
            tot = loss.detach() if tot is None else tot + loss.detach()
            terms.append(indiv)
        self.reducer.finish()
        grad_norm = self.optimizer.step(lr=self.learning_rate)
        self.iteration += 1
        self.learning_rate = update_learning_rate(self.hparams, self.iteration)
        return tot, terms, grad_norm

    def note_loss(self, value: float):
        """Host-side NaN bookkeeping for callers that fetch the loss (the reference skips logging, not the update)."""
        if math.isnan(value):
            self.nan_steps += 1

    # -- checkpoints: train.py:63-85 / :88-145 ---------------------------------------------------------------------------
    def checkpoint(self):
        cfg = {k: v for k, v in self.hparams.__dict__.items()}
        return {'iteration': self.iteration - 1, 'learning_rate': self.learning_rate, 'best_val_loss': self.best_val_loss,
                'state_dict': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'optimizer': self.optimizer.state_dict(), 'config_params': cfg}

    def save_checkpoint(self, filepath):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
        torch.save(self.checkpoint(), filepath)

    def load_checkpoint(self, checkpoint):
        """``checkpoint``: a path (read with ``weights_only=True``) or the dict itself.  Keys with DDP's ``module.`` prefix
        are accepted (train.py:272; consumers strip it, fine_tune.py:40)."""
        if isinstance(checkpoint, str):
            checkpoint = torch.load(checkpoint, map_location=self.device, weights_only=True)
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in checkpoint['state_dict'].items()}
        self.model.load_state_dict(sd)
        opt = checkpoint.get('optimizer')
        if opt is not None and len(opt['param_groups']) == len(self.optimizer.param_groups):
            self.optimizer.load_state_dict(opt)                          # else: keep the blank optimiser (train.py:127-133)
        self.iteration = int(checkpoint['iteration']) + 1                # "next iteration is iteration + 1" (train.py:289)
        self.learning_rate = update_learning_rate(self.hparams, self.iteration)   # recomputed from the schedule (train.py:294)
        self.best_val_loss = checkpoint.get('best_val_loss', float('inf'))
        self.model.runtime.invalidate_packs()
