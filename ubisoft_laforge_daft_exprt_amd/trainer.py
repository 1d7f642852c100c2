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
  * the backward runs in PHASES, cut where a finished part's gradient buckets can be handed to RCCL while the rest still computes
    (``cuts``; functional.cut): A = zero the gradient buckets, forward, loss, backward of everything downstream of the accent
    embedding (29 MB of gradients); B = backward of the accent encoder's four FFT blocks (13.7 MB); C = prenet layers 2 and 1
    (14.2 MB, 12.6 of them the 1024 x 1024 layer); D = prenet layer 0 (1 MB).  Each phase ends by launching its queued weight
    gradients (ops.flush_wgrads) and then its group's bucketed all-reduces on RCCL's stream: group g travels under phase g + 1, and
    only group D's 1 MB (2 % of the 57.5 MB) is launched after the last backward kernel (``exchange_plan()``).  With ONE rank there
    is nothing to exchange and the step is one phase (``cuts='auto'``).
  * the device work of each phase is ONE captured HIP graph per padded batch shape, replayed with the new batch copied into the
    static input buffers: ~600 kernel launches cost the host 7 ms per step when issued one by one from Python -- as much as the GPU
    needs to execute them -- and ~20 us per graph launch.  The same phase order runs eagerly (``use_graphs=False``).
  * what changes between replays lives in device memory the graphs read: the batch, the lengths, the adversarial-loss weight, the
    conditioner's speaker tables (updated in place, conditioning.BatchConditioner) and a dropout seed offset that the first graph
    increments (``runtime.seed_offset``), so every replay draws fresh dropout masks.  The MFMA weight packs are re-packed by the
    optimiser step, and at the top of ``train_step`` when anything else changed the parameters (``load_checkpoint``).
  * the optimiser (fused Adam, LR by value) and the one-launch weight re-pack stay outside the graphs: ten launches.
  * ``validate`` = train.py:163-209: eval-mode forward + loss under ``no_grad`` as one captured graph per padded shape.

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


def group_of(name: str, levels: int = 3) -> int:
    """Gradient-exchange group of a parameter = the backward PHASE after which its gradient is complete (``levels`` cuts, functional.cut):
    0 = everything downstream of the accent embedding, 1 = the accent encoder's four FFT blocks, 2 = prenet layers 1 and 2,
    3 = prenet layer 0 and the accent encoder's prosody embeddings (their gradients are complete earlier; they ride in the last, 1 MB
    group so that every group is ONE contiguous run of the reversed registration order)."""
    if levels < 1 or not name.startswith('accent_encoder.'):
        return 0
    if name.startswith('accent_encoder.blocks.'):
        return 1
    if name.startswith(('accent_encoder.convs.0.', 'accent_encoder.convs.2.', 'accent_encoder.energy_embedding.', 'accent_encoder.pitch_embedding.')):
        return min(levels, 3)
    return min(levels, 2)


PHASE_NAMES = ('A: forward, loss, backward of decoder / upsampler / phoneme encoder / style adapter / classifier',
               'B: backward of the accent encoder\'s four FFT blocks', 'C: backward of prenet layers 2 and 1 (+ prosody embeddings)',
               'D: backward of prenet layer 0')


class _StepGraphs:
    """The captured graphs (one per backward phase) of one padded batch shape with their static inputs and outputs."""
    __slots__ = ('graphs', 'inputs', 'loss', 'terms_dev', 'hits')


class Trainer:
    def __init__(self, model, criterion, hparams, conditioner=None, process_group=None, bucket_mb=16.0, grad_sink=True,
                 use_graphs=True, max_graphs=16, cuts='auto'):
        """``cuts``: how many times the backward is cut into phases (0..3) so that a finished phase's gradient buckets are exchanged
        while the next phase computes.  'auto' = 3 when there is more than one rank, 0 (one phase, one graph) otherwise."""
        self.model, self.criterion, self.hparams, self.conditioner = model, criterion, hparams, conditioner
        self.device = next(model.parameters()).device
        import torch.distributed as dist
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cut_levels = (3 if (world > 1 or os.environ.get('DX_FORCE_COLLECTIVES', '0') == '1') else 0) if cuts == 'auto' else int(cuts)
        if not 0 <= self.cut_levels <= 3:
            raise ValueError(f'cuts must be 0..3 or "auto", got {cuts!r}')
        self.reducer = GradientReducer(model, bucket_mb=bucket_mb, process_group=process_group, grad_sink=grad_sink,
                                       group_of=lambda n: group_of(n, self.cut_levels), explicit_launch=True)
        self.use_graphs, self.max_graphs = bool(use_graphs), int(max_graphs)
        # the FFT blocks' weight gradients go straight into the buckets (sink) and have no consumer before the exchange, so they are
        # queued during backward and launched 8 layers at a time (ops.flush_wgrads, after every backward phase below)
        model.runtime.defer_wgrad = bool(grad_sink) and os.environ.get('DX_DEFER_WGRAD', '1') != '0'
        # fp16 operand mode: gradients that live in 16-bit tensors (dqkv, the 1024-wide hidden gradients) would underflow; the
        # backward runs on loss * loss_scale and the fused Adam multiplies by 1 / loss_scale (static scale; bf16 needs none).
        # An overflow (non-finite gradient norm) SKIPS the update on every rank (the norm is computed after the all-reduce, so all
        # ranks see the same value): ``skipped_steps()`` counts them, ``hparams.loss_scale`` lowers the scale.
        self._fp16_loss_scale = float(getattr(hparams, 'loss_scale', 4096.0))
        self.graphs = {}
        self.val_graphs = {}
        self._conditioner_generation = getattr(conditioner, 'generation', None)
        self.adv_weight = torch.zeros((), dtype=torch.float32, device=self.device)    # read by the captured loss
        self.val_adv_weight = torch.zeros((), dtype=torch.float32, device=self.device)   # validation: iteration = 0 (train.py:198)
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        if model.runtime.seed_offset is None:
            model.runtime.seed_offset = torch.zeros((), dtype=torch.int64, device=self.device)
        self.optimizer = FusedAdam(self.reducer, lr=hparams.initial_learning_rate, betas=hparams.betas, eps=hparams.epsilon,
                                   weight_decay=hparams.weight_decay, grad_clip_thresh=hparams.grad_clip_thresh)
        self.accumulation_steps = int(getattr(hparams, 'accumulation_steps', 1))
        self.iteration = 1
        self.learning_rate = update_learning_rate(hparams, self.iteration)
        self.last_learning_rate = self.learning_rate      # the rate the most recent update used (what a checkpoint records, train.py:451-455)
        self.nan_steps = 0
        self.best_val_loss = float('inf')
        model.train()

    @property
    def loss_scale(self):
        return self._fp16_loss_scale if self.model.runtime.precision == 'fp16' else 1.0

    def skipped_steps(self) -> int:
        """Updates the fused Adam skipped because the gradient norm was not finite (one host sync)."""
        return self.optimizer.skipped_steps()

    def exchange_plan(self):
        """Per gradient-exchange group: its bytes and the point of the step at which its all-reduces are launched.  The LAST group is
        launched after the last backward kernel: its bytes are the exposed ones."""
        red = self.reducer
        plan = []
        for gid in sorted(set(red.bucket_group)):
            nbytes = 4 * sum(f.numel() for f, g in zip(red.flat, red.bucket_group) if g == gid)
            plan.append({'group': gid, 'bytes': nbytes, 'buckets': sum(1 for g in red.bucket_group if g == gid),
                         'launched_after_phase': PHASE_NAMES[gid]})
        total = sum(p['bytes'] for p in plan)
        return {'groups': plan, 'total_bytes': total, 'exposed_bytes': plan[-1]['bytes'], 'exposed_fraction': round(plan[-1]['bytes'] / total, 4)}

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
        """zero the buckets; per micro-batch: forward, loss, backward phase A (everything downstream of the accent embedding).
        ``launch(gid)`` is called when group ``gid``'s gradients are complete.  Returns (loss summed over micro-batches, list of
        LossTerms, list of callables running the remaining phases B, C, D in order -- each ends with its own ``launch``)."""
        model, red, k = self.model, self.reducer, self.accumulation_steps
        rt = model.runtime
        red.zero_grad()
        ops.begin_step_arena(rt, self.device)       # one zero fill each for the step's small accumulators
        ops.begin_step_arena(self.criterion.runtime, self.device)
        rt.backward_split, rt.cut_levels = [], self.cut_levels
        tot, terms = None, []
        # d(what is differentiated) / d(loss) = loss scale / accumulation steps is folded into the loss kernels' own gradient outputs
        self.criterion.grad_scale = self.loss_scale / k
        try:
            for inputs, targets in parsed:
                loss, indiv = self._forward_loss(inputs, targets, iteration)
                with red.accumulate(sync=False):                 # the exchange is launched explicitly, group by group
                    loss.backward(gradient=self._one)
                part = loss.detach() / k if k != 1 else loss.detach()
                tot = part if tot is None else tot + part
                terms.append(indiv)
        finally:
            cuts, rt.backward_split = rt.backward_split, None
            self.criterion.grad_scale = None
        ops.flush_wgrads(rt)                         # the FFT blocks' queued weight gradients, 8 layers per launch

        def end_arenas():
            ops.end_step_arena(rt)
            ops.end_step_arena(self.criterion.runtime)
        if self.cut_levels == 0:
            end_arenas()
        launch(0)

        def phase(level):
            def run():
                for lv, out, leaf in cuts:
                    if lv == level:
                        with red.accumulate(sync=False):
                            out.backward(leaf.grad)
                ops.flush_wgrads(rt)
                if level == self.cut_levels:
                    end_arenas()
                    cuts.clear()
                launch(level)
            return run
        return tot, terms, [phase(level) for level in range(1, self.cut_levels + 1)]

    def _parse(self, batches):
        parsed = [self.model.parse_batch(self.device, b) for b in batches]
        key = (self.model.runtime.precision,) + tuple((tuple(i[0].shape), tuple(i[8].shape)) for i, _ in parsed)   # (B, L_max), (B, n_mel, T_max) per micro-batch
        return parsed, key

    def _static_inputs(self, parsed):
        static = []
        for inputs, targets in parsed:
            si = tuple(t.clone() for t in inputs)
            for i in (5, 9):                                  # host lengths ride along: shapes / maxima are part of the key
                h = getattr(inputs[i], '_dx_host_lengths', None)
                si[i]._dx_host_lengths = h if h is not None else inputs[i].tolist()   # (a sync, at capture time only)
            st = (si[1], si[3], si[4], si[8], si[9], si[10])  # the targets alias the inputs, as parse_batch builds them
            static.append((si, st))
        return static

    def _capture(self, parsed, key):
        """One graph per backward phase for this padded shape.  One eager step on a side stream first (kernel attributes, weight
        packs, allocator), with the optimiser NOT applied: the captured step then sees the same state a replay will."""
        g = _StepGraphs()
        static = self._static_inputs(parsed)
        rt = self.model.runtime
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _, _, rest = self._phases(static, self.adv_weight, lambda gid: None)
            for run in rest:
                run()
        torch.cuda.current_stream().wait_stream(side)
        g.graphs = [torch.cuda.CUDAGraph() for _ in range(self.cut_levels + 1)]
        # thread_local: CUDA calls of OTHER threads (the RCCL watchdog polling its events, a pinned-memory loader) must not abort the capture
        with torch.cuda.graph(g.graphs[0], capture_error_mode='thread_local'):
            rt.seed_offset.add_(1)                            # a new dropout stream per replay (the seeds in the launches are frozen)
            g.loss, terms, rest = self._phases(static, self.adv_weight, lambda gid: None)
        g.terms_dev = [t._device_terms for t in terms]        # static device tensors: wrapped anew after every replay
        for graph, run in zip(g.graphs[1:], rest):
            with torch.cuda.graph(graph, pool=g.graphs[0].pool(), capture_error_mode='thread_local'):
                run()
        g.inputs, g.hits = static, 0
        if len(self.graphs) >= self.max_graphs:               # evict the least used shape (its pool is freed with it)
            del self.graphs[min(self.graphs, key=lambda q: self.graphs[q].hits)]
        self.graphs[key] = g
        return g

    def _check_conditioner(self):
        """A conditioner that had to RE-ALLOCATE its device tables (a speaker id beyond their rows) invalidates every captured graph:
        the graphs hold the old pointers and row count.  In-place refreshes (the normal case) need nothing."""
        gen = getattr(self.conditioner, 'generation', None)
        if gen != self._conditioner_generation:
            self.graphs.clear()
            self.val_graphs.clear()
            self._conditioner_generation = gen

    def resident_batch(self, batch):
        """The batch copied into the static input buffers of its shape's graphs (captured now if need be) and handed back as a
        reference 14-tuple of THOSE tensors: ``train_step`` then finds its inputs in place (a data loader writing pinned host batches
        straight into the static buffers does the same).  One micro-batch per update only."""
        if self.accumulation_steps != 1 or not self.use_graphs:
            return batch
        self._check_conditioner()
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
        # a replayed graph reads the MFMA weight packs through frozen pointers and never runs the Python staleness check: anything that
        # changed the parameters since the last optimiser step (load_checkpoint, load_state_dict, an in-place edit) is re-packed here.
        # One launch when something is stale, nothing otherwise.
        ops.repack_all(self.model.runtime)
        if self.use_graphs:
            self._check_conditioner()
            self.adv_weight.fill_(self.criterion.update_adversarial_weight(self.iteration))
            g = self.graphs.get(key)
            if g is None:
                g = self._capture(parsed, key)
            g.hits += 1
            for (si, _), (inputs, _) in zip(g.inputs, parsed):
                for dst, src in zip(si, inputs):
                    if dst.data_ptr() != src.data_ptr():      # a batch built by resident_batch() already lives in the static buffers
                        dst.copy_(src, non_blocking=True)
            for gid, graph in enumerate(g.graphs):
                graph.replay()
                red.launch_group(gid)                         # exchanged (RCCL, its own stream) while the next phase's graph runs
            from .loss import LossTerms
            tot, terms = g.loss, [LossTerms(t) for t in g.terms_dev]
        else:
            tot, terms, rest = self._phases(parsed, self.iteration, red.launch_group)
            for run in rest:
                run()
        red.finish()
        grad_norm = self.optimizer.step(lr=self.learning_rate, grad_scale=1.0 / self.loss_scale)
        self.last_learning_rate = self.learning_rate
        self.iteration += 1
        self.learning_rate = update_learning_rate(self.hparams, self.iteration)
        return tot, terms, grad_norm

    # -- validation: train.py:163-209 ---------------------------------------------------------------------------------------
    def _val_forward(self, inputs, targets):
        with torch.no_grad():
            return self._forward_loss(inputs, targets, self.val_adv_weight)

    def validate(self, batches, keep_outputs=False):
        """The reference's ``validate`` (train.py:163-209, called at :474-491) on this rank's validation batches: eval mode (dropout
        off), no gradients, the same batch conditioning as training, ``iteration = 0`` in the loss (adversarial weight 0), losses
        averaged over the batches; every rank runs it, nothing is exchanged.  The device work of a batch is ONE captured graph per
        padded shape (forward + loss); the per-batch losses are summed on the device and fetched with one transfer at the end.
        Returns (val_loss, dict of the 7 averaged terms[, list of (targets, outputs) when ``keep_outputs``])."""
        from .loss import LossTerms
        model = self.model
        was_training = model.training
        model.eval()
        self._check_conditioner()
        ops.repack_all(model.runtime)
        tot = torch.zeros((), dtype=torch.float32, device=self.device)
        terms = torch.zeros(len(LossTerms.KEYS), dtype=torch.float32, device=self.device)
        kept, n = [], 0
        try:
            for batch in batches:
                (inputs, targets), = [model.parse_batch(self.device, batch)]
                if keep_outputs or not self.use_graphs:      # the caller wants the tensors themselves: eager
                    raw = (inputs[6], inputs[7])
                    cin = self.conditioner.process_batch(inputs, self.device) if self.conditioner is not None else inputs
                    tg = (targets[0], cin[3], cin[4], targets[3], targets[4], targets[5], raw[0], raw[1])
                    with torch.no_grad():
                        outputs = model(cin)
                        loss, indiv = self.criterion(outputs, tg, self.val_adv_weight)
                    if keep_outputs:
                        kept.append((tg, outputs))
                    tot += loss
                    terms += indiv._device_terms
                else:
                    key = (model.runtime.precision, tuple(inputs[0].shape), tuple(inputs[8].shape))
                    g = self.val_graphs.get(key)
                    if g is None:
                        g = self._capture_val([(inputs, targets)], key)
                    for dst, src in zip(g.inputs[0][0], inputs):
                        dst.copy_(src, non_blocking=True)
                    g.graphs[0].replay()
                    tot += g.loss
                    terms += g.terms_dev[0]
                n += 1
        finally:
            model.train(was_training)
        if n == 0:
            raise ValueError('validate() needs at least one batch')
        host = torch.cat([tot.reshape(1), terms]).div_(n).tolist()      # the one device -> host transfer of the validation pass
        out = (host[0], dict(zip(LossTerms.KEYS, host[1:])))
        return out + (kept,) if keep_outputs else out

    def _capture_val(self, parsed, key):
        g = _StepGraphs()
        static = self._static_inputs(parsed)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._val_forward(*static[0])                        # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        g.graphs = [torch.cuda.CUDAGraph()]
        with torch.cuda.graph(g.graphs[0], capture_error_mode='thread_local'):
            g.loss, indiv = self._val_forward(*static[0])
        g.terms_dev = [indiv._device_terms]
        g.inputs, g.hits = static, 0
        if len(self.val_graphs) >= self.max_graphs:
            self.val_graphs.pop(next(iter(self.val_graphs)))
        self.val_graphs[key] = g
        return g

    def note_loss(self, value: float):
        """Host-side NaN bookkeeping for callers that fetch the loss (the reference skips logging, not the update)."""
        if math.isnan(value):
            self.nan_steps += 1

    # -- checkpoints: train.py:63-85 / :88-145 ---------------------------------------------------------------------------
    def checkpoint(self):
        cfg = {k: v for k, v in self.hparams.__dict__.items()}
        return {'iteration': self.iteration - 1, 'learning_rate': self.last_learning_rate, 'best_val_loss': self.best_val_loss,
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
