"""The reference's optimiser-step structure around the HIP path ("boundary only" -- the trainer file itself
never travels, but a drop-in must reproduce what it does between two parameter updates).

One ``Trainer.train_step`` = src/daft_exprt/train.py:390-539 without logging / validation / checkpoint cadence:

    for each of ``accumulation_steps`` micro-batches:                       train.py:390-436
        parse_batch -> keep RAW frame energy / pitch for the consistency losses -> on-device batch conditioning
        -> model forward -> DaftExprtLoss(outputs, targets, iteration) -> (loss / accumulation_steps).backward()
    gradient exchange (bucketed RCCL all-reduce, launched during the LAST micro-batch's backward)        train.py:272 (DDP)
    clip_grad_norm_(parameters, grad_clip_thresh) + Adam(lr(iteration)) as ONE fused launch per bucket  train.py:443-445
    iteration += 1; lr = update_learning_rate(hparams, iteration)                                        train.py:517-539

MI355X-specific structure of one update (``use_graphs=True``, the default):
  * the device work of a step is TWO captured HIP graphs per padded batch shape, replayed with the new batch copied into their
    static input buffers: graph A = zero the gradient buckets, forward, loss, backward of everything downstream of the accent
    embedding; graph B = backward of the accent encoder (~45 % of the backward).  ~600 kernel launches cost the host 7 ms per
    step when issued one by one from Python -- as much as the GPU needs to execute them -- and ~20 us as two graph launches.
  * the cut is also where the gradient exchange overlaps: the buckets of graph A's parameters are all-reduced (RCCL, its own
    stream) while graph B runs; only the accent encoder's buckets are exchanged after it.  The same two-phase order runs eagerly
    (``use_graphs=False``), and on one GPU.
  * what changes between replays lives in device memory the graphs read: the batch, the lengths, the adversarial-loss weight, and
    a dropout seed offset that graph A increments (``runtime.seed_offset``), so every replay draws fresh dropout masks.
  * the optimiser (fused Adam, LR by value) and the one-launch weight re-pack stay outside the graphs: ten launches.

Like the reference, a NaN loss does not stop the update (train.py:445-450 only skips LOGGING); ``nan_steps`` counts them.
The iteration counter starts at 1 (train.py:286) and the learning rate of step ``i`` is ``update_learning_rate(hparams, i)``.
``save_checkpoint`` / ``load_checkpoint`` use the reference's dict layout (train.py:63-145) including the torch.optim.Adam
optimiser layout.
"""
from __future__ import annotations

import math
import os

import torch

from . import ops
from .ddp import GradientReducer
from .optim import FusedAdam, update_learning_rate


def _group_of(name: str) -> int:
    """gradient-exchange groups = the two backward phases: 1 = accent encoder (phase B), 0 = everything else (phase A)"""
    return 1 if name.startswith('accent_encoder.') else 0


class _StepGraphs:
    """The two captured graphs of one padded batch shape with their static inputs and outputs."""
    __slots__ = ('graph_a', 'graph_b', 'inputs', 'loss', 'terms_dev', 'hits')


class Trainer:
    def __init__(self, model, criterion, hparams, conditioner=None, process_group=None, bucket_mb=16.0, grad_sink=True,
                 use_graphs=True, max_graphs=16):
        self.model, self.criterion, self.hparams, self.conditioner = model, criterion, hparams, conditioner
        self.device = next(model.parameters()).device
        self.reducer = GradientReducer(model, bucket_mb=bucket_mb, process_group=process_group, grad_sink=grad_sink, group_of=_group_of)
        self.use_graphs, self.max_graphs = bool(use_graphs), int(max_graphs)
        # the FFT blocks' weight gradients go straight into the buckets (sink) and have no consumer before the exchange, so they are
        # queued during backward and launched 8 layers at a time (ops.flush_wgrads, after every backward phase below)
        model.runtime.defer_wgrad = bool(grad_sink) and os.environ.get('DX_DEFER_WGRAD', '1') != '0'
        # fp16 operand mode: gradients that live in 16-bit tensors (dqkv, the 1024-wide hidden gradients) would underflow; the
        # backward runs on loss * loss_scale and the fused Adam multiplies by 1 / loss_scale (static scale; bf16 needs none)
        self._fp16_loss_scale = float(getattr(hparams, 'loss_scale', 4096.0))
        self.graphs = {}
        self.adv_weight = torch.zeros((), dtype=torch.float32, device=self.device)    # read by the captured loss
        if model.runtime.seed_offset is None:
            model.runtime.seed_offset = torch.zeros((), dtype=torch.int64, device=self.device)
        self.optimizer = FusedAdam(self.reducer, lr=hparams.initial_learning_rate, betas=hparams.betas, eps=hparams.epsilon,
                                   weight_decay=hparams.weight_decay, grad_clip_thresh=hparams.grad_clip_thresh)
        self.accumulation_steps = int(getattr(hparams, 'accumulation_steps', 1))
        self.iteration = 1
        self.learning_rate = update_learning_rate(hparams, self.iteration)
        self.nan_steps = 0
        self.best_val_loss = float('inf')
        model.train()

    @property
    def loss_scale(self):
        return self._fp16_loss_scale if self.model.runtime.precision == 'fp16' else 1.0

    # -- device work of one update ----------------------------------------------------------------------------------------
    def _forward_loss(self, inputs, targets, iteration):
        """train.py:405-422 on parsed device tensors: RAW frame prosody for the consistency losses, conditioning, forward, loss."""
        raw_frames_energy, raw_frames_pitch = inputs[6], inputs[7]       # the frozen pitch predictor outputs RAW pitch
        if self.conditioner is not None:
            inputs = self.conditioner.process_batch(inputs, self.device)   # length tensors pass through (host lengths ride along)
        targets = (targets[0], inputs[3], inputs[4], targets[3], targets[4], targets[5], raw_frames_energy, raw_frames_pitch)
        outputs = self.model(inputs)
        return self.criterion(outputs, targets, iteration)

    def _phases(self, parsed, iteration, launch):
        """zero the buckets; per micro-batch: forward, loss, backward phase A (everything downstream of the accent embedding), then
        phase B (the accent encoder).  ``launch(gid)`` is called when group ``gid``'s gradients are complete.
        Returns (loss summed over micro-batches, list of LossTerms, callable running phase B)."""
        model, red, k = self.model, self.reducer, self.accumulation_steps
        red.zero_grad()
        ops.begin_step_arena(model.runtime, self.device)       # one zero fill each for the step's small accumulators
        ops.begin_step_arena(self.criterion.runtime, self.device)
        model.backward_split = []
        tot, terms = None, []
        for inputs, targets in parsed:
            loss, indiv = self._forward_loss(inputs, targets, iteration)
            with red.accumulate(sync=False):                 # the exchange is launched explicitly, group by group
                (loss * (self.loss_scale / k)).backward()
            ops.flush_wgrads(model.runtime)                  # the FFT blocks' queued weight gradients, 8 layers per launch
            part = loss.detach() / k if k != 1 else loss.detach()
            tot = part if tot is None else tot + part
            terms.append(indiv)
        cuts, model.backward_split = model.backward_split, None
        launch(0)

        def phase_b():
            for emb, leaf in cuts:
                with red.accumulate(sync=False):
                    emb.backward(leaf.grad)
                ops.flush_wgrads(model.runtime)
            ops.end_step_arena(model.runtime)
            ops.end_step_arena(self.criterion.runtime)
            launch(1)
        return tot, terms, phase_b

    def _parse(self, batches):
        parsed = [self.model.parse_batch(self.device, b) for b in batches]
        key = (self.model.runtime.precision,) + tuple((tuple(i[0].shape), tuple(i[8].shape)) for i, _ in parsed)   # (B, L_max), (B, n_mel, T_max) per micro-batch
        return parsed, key

    def _capture(self, parsed, key):
        """Two graphs for this padded shape.  One eager step on a side stream first (kernel attributes, weight packs, allocator),
        with the optimiser NOT applied: the captured step then sees the same state a replay will."""
        g = _StepGraphs()
        static = []
        for inputs, targets in parsed:
            si = tuple(t.clone() for t in inputs)
            for i in (5, 9):                                  # host lengths ride along: shapes / maxima are part of the key
                h = getattr(inputs[i], '_dx_host_lengths', None)
                si[i]._dx_host_lengths = h if h is not None else inputs[i].tolist()   # (a sync, at capture time only)
            st = (si[1], si[3], si[4], si[8], si[9], si[10])  # the targets alias the inputs, as parse_batch builds them
            static.append((si, st))
        rt = self.model.runtime
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _, _, pb = self._phases(static, self.adv_weight, lambda gid: None)
            pb()
        torch.cuda.current_stream().wait_stream(side)
        g.graph_a, g.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: CUDA calls of OTHER threads (the RCCL watchdog polling its events, a pinned-memory loader) must not abort the capture
        with torch.cuda.graph(g.graph_a, capture_error_mode='thread_local'):
            rt.seed_offset.add_(1)                            # a new dropout stream per replay (the seeds in the launches are frozen)
            g.loss, terms, phase_b = self._phases(static, self.adv_weight, lambda gid: None)
        g.terms_dev = [t._device_terms for t in terms]        # static device tensors: wrapped anew after every replay
        with torch.cuda.graph(g.graph_b, pool=g.graph_a.pool(), capture_error_mode='thread_local'):
            phase_b()
        g.inputs, g.hits = static, 0
        if len(self.graphs) >= self.max_graphs:               # evict the least used shape (its pool is freed with it)
            del self.graphs[min(self.graphs, key=lambda q: self.graphs[q].hits)]
        self.graphs[key] = g
        return g

    def resident_batch(self, batch):
        """The batch copied into the static input buffers of its shape's graphs (captured now if need be) and handed back as a
        reference 14-tuple of THOSE tensors: ``train_step`` then finds its inputs in place (a data loader writing pinned host batches
        straight into the static buffers does the same).  One micro-batch per update only."""
        if self.accumulation_steps != 1 or not self.use_graphs:
            return batch
        parsed, key = self._parse([batch])
        g = self.graphs.get(key) or self._capture(parsed, key)
        si = g.inputs[0][0]
        for dst, src in zip(si, parsed[0][0]):
            dst.copy_(src)
        return (si[0], si[1], si[2], si[3], si[4], si[5], si[6], si[7], si[8], si[9], si[10], batch[11], batch[12], si[11])

    def train_step(self, batches):
        """``batches``: the ``accumulation_steps`` micro-batches (reference 14-tuples) of one parameter update.
        Returns (loss tensor = mean over micro-batches, list of per-micro-batch LossTerms, gradient norm tensor)."""
        batches = list(batches)
        if len(batches) != self.accumulation_steps:
            raise ValueError(f'expected {self.accumulation_steps} micro-batches, got {len(batches)}')
        parsed, key = self._parse(batches)
        red = self.reducer
        if self.use_graphs:
            self.adv_weight.fill_(self.criterion.update_adversarial_weight(self.iteration))
            g = self.graphs.get(key)
            if g is None:
                g = self._capture(parsed, key)
            g.hits += 1
            for (si, _), (inputs, _) in zip(g.inputs, parsed):
                for dst, src in zip(si, inputs):
                    if dst.data_ptr() != src.data_ptr():      # a batch built by resident_batch() already lives in the static buffers
                        dst.copy_(src, non_blocking=True)
            g.graph_a.replay()
            red.launch_group(0)                               # exchanged while graph B (accent-encoder backward) runs
            g.graph_b.replay()
            red.launch_group(1)
            from .loss import LossTerms
            tot, terms = g.loss, [LossTerms(t) for t in g.terms_dev]
        else:
            tot, terms, phase_b = self._phases(parsed, self.iteration, red.launch_group)
            phase_b()
        red.finish()
        grad_norm = self.optimizer.step(lr=self.learning_rate, grad_scale=1.0 / self.loss_scale)
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
