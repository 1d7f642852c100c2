"""Data-parallel gradient exchange: bucketed all-reduce over RCCL (xGMI), overlapped with backward.

The reference wraps the model in ``torch.nn.parallel.DistributedDataParallel`` (src/daft_exprt/train.py:272); the only
exchange step of the path is the gradient all-reduce of 14.4 M fp32 values (57.5 MB) per step.
Utterances are independent units, so the forward/backward itself needs no collective.

Design for one process per GPU on an 8-GPU xGMI mesh:
  * like DistributedDataParallel, construction broadcasts rank 0's parameters (and buffers), so replicas that were initialised
    with different seeds start identical
  * parameters are packed, in reverse registration order (~ the order autograd finishes them), into a few flat fp32
    buckets; ``param.grad`` is a VIEW into its bucket, so autograd accumulates straight into the communication buffer
  * a post-accumulate hook counts ready parameters; when a bucket is complete its all-reduce is issued asynchronously
    (``backend='nccl'`` is RCCL on ROCm) while the rest of backward keeps running on the compute stream
  * ``finish()`` waits for the outstanding collectives; averaging uses ReduceOp.AVG where the backend has it
  * gradient accumulation over micro-batches (train.py:421-441): ``with reducer.accumulate(sync=is_last):`` around each
    micro-batch's backward; only the last one launches collectives (one exchange of the accumulated sum instead of the
    reference's one per micro-batch -- the average is the same)
Few, large messages: with 7 point-to-point links per GPU the all-reduce is per-link bound, so 2-4 buckets of 16-32 MB
amortise latency without delaying the first launch until the end of backward.
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


class GradientReducer:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 16.0, process_group=None, grad_sink: bool = False,
                 broadcast_parameters: bool = True, group_of=None, explicit_launch: bool = False):
        """``grad_sink=True``: the HIP backward kernels accumulate straight into the bucket views (the model's ``runtime.sink``);
        requires ``zero_grad()`` of THIS object before every step.
        ``group_of(parameter name) -> int``: parameters of different groups never share a bucket (``bucket_group[i]`` is bucket
        i's group), so a caller that finishes backward group by group can exchange a group as soon as it is complete.
        ``explicit_launch=True`` (trainer.Trainer): the caller runs the backward in phases and launches every group itself
        (``launch_group``); the hooks then only COUNT gradients -- a bucket may fill over several phases and micro-batches -- and
        ``finish()`` checks that every parameter was counted equally often."""
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # DX_FORCE_COLLECTIVES=1 (rehearsal on one GPU with a one-rank RCCL group): issue the all-reduces although there is nobody to exchange with
        import os
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get('DX_FORCE_COLLECTIVES', '0') == '1')
        if self.world > 1 and broadcast_parameters:
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=src, group=process_group)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        named.reverse()
        params = [p for _, p in named]
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, self.bucket_group = [], []
        cur, cur_n, cur_g = [], 0, None
        for n, p in named:
            gid = group_of(n) if group_of is not None else 0
            if cur and (cur_n + p.numel() > cap or gid != cur_g):
                self.buckets.append(cur)
                self.bucket_group.append(cur_g)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
            cur_g = gid
        if cur:
            self.buckets.append(cur)
            self.bucket_group.append(cur_g)
        self.flat, self.bucket_of = [], {}
        # ONE allocation for all buckets: zeroing, the gradient norm and the fused Adam step are then one launch each over the whole range
        # instead of one per bucket; the gaps between buckets stay zero.  Every bucket starts on a 4 KB boundary: a bucket under exchange
        # (the collective's loads / stores, on its own queue) then shares no cache line or page with a bucket whose gradients are still
        # receiving the backward kernels' memory-side float atomics.
        sizes = [sum(p.numel() for p in bucket) for bucket in self.buckets]
        self.bucket_offset, total = [], 0
        for n in sizes:
            self.bucket_offset.append(total)
            total += (n + 1023) // 1024 * 1024
        self.flat_all = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        for bi, bucket in enumerate(self.buckets):
            n = sizes[bi]
            flat = self.flat_all[self.bucket_offset[bi]:self.bucket_offset[bi] + n]
            off = 0
            for p in bucket:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[p] = bi
            self.flat.append(flat)
        self.pending = [len(b) for b in self.buckets]
        self.explicit = bool(explicit_launch)
        self.seen = [0] * len(self.buckets)       # explicit mode: gradients counted per bucket since zero_grad()
        self.launched = set()                     # buckets whose exchange has been launched explicitly in this step
        self.works = []
        self.sync = True
        self.hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self._avg = None
        self.runtime = getattr(module, 'runtime', None) if grad_sink else None
        if grad_sink:
            if self.runtime is None:
                raise ValueError('grad_sink=True needs a module with a ``runtime`` (ubisoft_laforge_daft_exprt_amd.DaftExprt)')
            self.runtime.sink = True

    def _reduce_op(self):
        if self._avg is None:
            backend = dist.get_backend(self.group)
            self._avg = dist.ReduceOp.AVG if backend == 'nccl' else dist.ReduceOp.SUM
        return self._avg

    def _on_grad(self, p):
        bi = self.bucket_of[p]
        if self.explicit:
            self.seen[bi] += 1
            return
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and self.collective and self.sync:
            op = self._reduce_op()
            work = dist.all_reduce(self.flat[bi], op=op, group=self.group, async_op=True)
            self.works.append((work, bi, op))

    def zero_grad(self):
        """Keeps the grad views alive (set_to_none would detach them from the buckets)."""
        self.flat_all.zero_()
        self.launched.clear()
        self.pending = [len(b) for b in self.buckets]
        self.seen = [0] * len(self.buckets)

    @contextlib.contextmanager
    def accumulate(self, sync: bool = True):
        """One micro-batch's backward inside a gradient-accumulation step.  ``sync=False``: its gradients are only summed
        into the buckets; ``sync=True`` (the last micro-batch): completed buckets are all-reduced as usual."""
        if self.explicit:                          # phases of a trainer's backward: nothing to reset, nothing launches by itself
            yield self
            return
        self._check_balanced()
        self.pending = [0 if bi in self.launched else len(b) for bi, b in enumerate(self.buckets)]
        self.sync = bool(sync)
        try:
            yield self
        finally:
            self.sync = True

    def _check_balanced(self):
        if any(n != 0 and n != len(b) for n, b in zip(self.pending, self.buckets)):
            raise RuntimeError(f'gradient bookkeeping out of balance: pending per bucket = {self.pending} '
                               '(a parameter received no gradient, or was counted twice)')

    def finish(self):
        """Call after backward: waits for the collectives; gradients are then averaged over ranks."""
        if self.explicit:
            rounds = {n / len(b) for n, b in zip(self.seen, self.buckets)}
            # (a replayed HIP graph runs no Python hooks: all counts are then zero, and the same sequence was checked when it was captured)
            if len(rounds) != 1 or min(rounds) != int(min(rounds)) or len(self.launched) != len(self.buckets):
                raise RuntimeError(f'gradient bookkeeping out of balance after backward: gradients counted per bucket = {self.seen} for buckets of '
                                   f'{[len(b) for b in self.buckets]} parameters, launched = {sorted(self.launched)} (a parameter received no '
                                   'gradient, was counted twice, or a group was never launched)')
            self.seen = [0] * len(self.buckets)
        elif any(n != 0 for n in self.pending):
            raise RuntimeError(f'gradient bookkeeping out of balance after backward: pending per bucket = {self.pending} '
                               '(a parameter received no gradient, or was counted twice)')
        for work, bi, op in self.works:
            work.wait()
            if op == dist.ReduceOp.SUM and self.world > 1:
                self.flat[bi].div_(self.world)
        self.works = []
        self.launched.clear()
        self.pending = [len(b) for b in self.buckets]

    def launch_group(self, gid=None):
        """Explicit asynchronous exchange of the buckets of group ``gid`` (all buckets if None): for callers whose backward does not
        run Python hooks -- a replayed HIP graph.  The collectives run on RCCL's stream behind everything enqueued on the compute
        stream so far; ``finish()`` waits for them."""
        for bi, flat in enumerate(self.flat):
            if gid is None or self.bucket_group[bi] == gid:
                self.pending[bi] = 0
                self.launched.add(bi)
                if self.collective:
                    op = self._reduce_op()
                    self.works.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), bi, op))

    def remove(self):
        for h in self.hooks:
            h.remove()
        if self.runtime is not None:
            self.runtime.sink = False
