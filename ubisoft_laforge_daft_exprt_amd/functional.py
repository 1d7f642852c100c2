"""Autograd glue: one ``torch.autograd.Function`` per fused block of the reference model, each a fixed sequence of
C-ABI kernel launches (see ops.py).  Saved activations are the kernels' own outputs; dropout masks are never stored
(the backward kernels regenerate them from the same counter-based seed).
"""
from __future__ import annotations

import threading

import torch

from . import ops

# ----------------------------------------------------------------------------------------------------------------------
# dropout seeds: every dropout site of every forward call draws a fresh 64-bit stream id
# ----------------------------------------------------------------------------------------------------------------------
_seed_state = threading.local()


def manual_seed(seed: int) -> None:
    _seed_state.base = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF
    _seed_state.counter = 0


def next_seed() -> int:
    if not hasattr(_seed_state, 'base'):
        manual_seed(torch.initial_seed())
    _seed_state.counter += 1
    return (_seed_state.base + _seed_state.counter * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------------------------------------------------
# gradient sink: when a trainer pre-allocates (and zeroes) ``param.grad`` -- as ddp.GradientReducer does with views into its
# communication buckets -- the backward kernels accumulate straight into it (atomics / accumulate epilogues) and hand
# autograd ``None``: no temporary gradient tensor, no ATen ``+=`` launch per parameter.  The autograd engine still runs the
# parameter's post-accumulate hooks for an undefined gradient (torch 2.10), which is what drives the reducer's bucket
# bookkeeping; GradientReducer.finish() raises if that ever stops being true.
# The switch lives on the model's ops.Runtime (``rt.sink``); every Function captures it in ``ctx`` at forward time.
# ----------------------------------------------------------------------------------------------------------------------
def _sink(param, enabled):
    if not enabled or param is None:
        return None
    g = param.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32:
        return None
    return g


class Lengths:
    """Valid lengths of one axis of a padded batch: int32 on the device for the kernels, Python ints for shapes."""

    def __init__(self, lengths: torch.Tensor, host=None):
        if not lengths.is_cuda:
            raise RuntimeError('lengths must live on the GPU (the kernels read them as device memory); got a CPU tensor')
        self.host = [int(v) for v in (lengths.tolist() if host is None else host)]  # one D2H sync unless given
        self.max = max(self.host)
        self.total = sum(self.host)
        self.i32 = lengths.to(dtype=torch.int32).contiguous()
        self.i64 = lengths
        # optional device int32 [B]: rows n >= exist[b] of utterance b do not exist for the k = 3 convolutions (they read zero), although
        # the tensors have more rows.  None = the reference's padded grid (every utterance has max(lengths) rows).  Set by
        # inference.GraphedSynthesizer: padded-shape buckets (exist = the batch's true longest length) and the batched accent encoder
        # (exist = lengths: every reference behaves as if it were run alone, scripts/synthesize.py:420-448).
        self.exist = None
        self._order = None

    @property
    def order(self):
        """device int32 [B]: utterance indices, longest first (ops.length_order) -- computed by one tiny launch on first use (under graph
        capture: recorded, so every replay re-derives it from the static length buffer).  Attention workgroups are handed out in this order."""
        if not ops._ATTN_ORDER:
            return None
        if self._order is None:
            self._order = ops.length_order(self.i32)
        return self._order


# ----------------------------------------------------------------------------------------------------------------------
# Linear (LinearNorm, model.py:57-72) on (rows, Cin)
# ----------------------------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pack, relu, grad_scale, lens, need_dx):
        x2 = x.reshape(-1, x.shape[-1]) if lens is None else x
        ctx.prec = pack.rt.precision                       # captured: the backward runs in the mode of ITS forward
        y = ops.conv_gemm(x2, pack, bias, relu=relu, lens=lens, mask_rows=lens is not None, prec=ctx.prec)
        ctx.save_for_backward(x2, y if relu else None)
        ctx.pack, ctx.relu, ctx.grad_scale, ctx.lens, ctx.need_dx, ctx.xshape = pack, relu, grad_scale, lens, need_dx, x.shape
        ctx.params, ctx.sink = (weight, bias), bool(pack.rt.sink)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        dy = dy.contiguous().view(*x2.shape[:-1], dy.shape[-1])
        if ctx.lens is not None:
            dy = ops.mask_rows(dy, ctx.lens)
        if ctx.relu:
            dy = ops.relu_bwd(dy, y)
        # gradient sink: the weight-gradient launch adds straight into the bucket views (no zero fill, no AccumulateGrad add per parameter)
        # (utterance-level linears - classifier, FiLM predictors, projections - are 48-row products of ~9 us each as launches of their own: queued,
        # they ride in ONE batched launch with the other k = 1 layers of their operand kind at the end of the backward phase)
        dw, db = ops.conv_wgrad(dy, x2, ctx.pack, arena=ctx.pack.rt.arena, w_sink=_sink(ctx.params[0], ctx.sink), b_sink=_sink(ctx.params[1], ctx.sink),
                                prec=ctx.prec, defer=ctx.lens is None)
        dx = None
        if ctx.need_dx:
            dx = ops.conv_gemm(dy, ctx.pack, None, transpose=True, out_scale=ctx.grad_scale, prec=ctx.prec).view(ctx.xshape)
        return dx, dw, db, None, None, None, None, None


class PaddedLinear:
    """Linear whose output width is not a multiple of 4 (speaker logits): the kernels see a zero-padded (Cout4, Cin) weight."""

    def __init__(self, weight, bias, rt=None):
        self.weight, self.bias = weight, bias
        self.rt = rt or ops.DEFAULT
        self.cout = weight.shape[0]
        self.cout4 = (self.cout + 3) // 4 * 4
        self._key = None
        self.wpad = self.bpad = self.pack = None

    def padded(self):
        # The key must see every way the parameter can change: autograd-visible writes (``_version``), re-homing of ``.data``
        # (``data_ptr``: optim.FusedAdam moves parameters into flat buckets) and raw-pointer writes by ``dx_adam_step`` (the pack
        # epoch, which the optimiser bumps after every step).
        key = (self.weight._version, self.bias._version, self.weight.data_ptr(), self.bias.data_ptr(), self.rt.pack_epoch)
        # under HIP-graph capture the refresh is recorded unconditionally: this Python check does not run on replay, and the
        # optimiser rewrites the parameter between replays
        if key != self._key or (self.wpad is not None and self.wpad.is_cuda and torch.cuda.is_current_stream_capturing()):
            w = self.weight.detach()
            if self.wpad is None or self.wpad.device != w.device:
                self.wpad = torch.zeros(self.cout4, w.shape[1], dtype=w.dtype, device=w.device)
                self.bpad = torch.zeros(self.cout4, dtype=w.dtype, device=w.device)
                self.pack = ops.PackedWeight(self.wpad, self.rt)
            self.wpad[:self.cout].copy_(w)                 # bumps wpad._version: the MFMA pack refreshes on its next use
            self.bpad[:self.cout].copy_(self.bias.detach())
            self._key = key
        return self.wpad, self.bpad, self.pack


class PaddedLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, padded: PaddedLinear):
        _, bpad, pack = padded.padded()
        ctx.prec = pack.rt.precision
        y = ops.conv_gemm(x, pack, bpad, prec=ctx.prec)
        ctx.save_for_backward(x)
        ctx.pack, ctx.cout = pack, padded.cout
        return y[:, :padded.cout].contiguous()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dyp = ops._zeros(ctx.pack.rt.arena, dy.shape[0], ctx.pack.cout, device=dy.device)
        dyp[:, :ctx.cout].copy_(dy)
        dw, db = ops.conv_wgrad(dyp, x, ctx.pack, arena=ctx.pack.rt.arena, prec=ctx.prec)
        dw, db = dw[:ctx.cout], db[:ctx.cout]
        dx = ops.conv_gemm(dyp, ctx.pack, None, transpose=True, prec=ctx.prec)
        return dx, dw, db, None


# ----------------------------------------------------------------------------------------------------------------------
# FFT block (model.py:238-259): MHA + dropout + residual LN + mask, conv FF + dropout + residual LN + FiLM + mask
# ----------------------------------------------------------------------------------------------------------------------
class FFTBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, film, lens, packs, cfg, training,
                in_w, in_b, out_w, out_b, ln1_w, ln1_b, c1_w, c1_b, c2_w, c2_b, ln2_w, ln2_b, qkv_pre=None, next_in=None):
        """-> (y2, qkv_next).  ``qkv_pre``: this block's q/k/v, already produced by the previous block's last launch; ``next_in`` = (pack, bias)
        of the next block's in-projection: where the fused feed-forward launch runs, it produces that block's q/k/v too (else None)."""
        p_attn = cfg['attn_dropout'] if training else 0.0
        p_conv = cfg['conv_dropout'] if training else 0.0
        heads = cfg['attn_nb_heads']
        s_attn, s_ln1, s_ln2 = (next_seed(), next_seed(), next_seed()) if training else (0, 0, 0)
        film = film if film is None or film.stride(-1) == 1 else film.contiguous()
        L = lens.i32
        rt = packs['in'].rt
        prec = rt.precision                        # captured here, used by the backward (never re-read)
        hd = ops.hidden_dtype(prec)
        qkv = qkv_pre if qkv_pre is not None else \
            ops.conv_gemm(x, packs['in'], in_b, lens=L, halo=0, out_dtype=hd, prec=prec)     # bf16 mode: attention reads bf16 q/k/v
        so = rt.seed_offset                        # device scalar added to the seeds (graph replays), or None
        sh = ops.gemm_shadow(prec)                 # bf16 mode: GEMM operands also exist as bf16 copies written by their producers
        if ops.attn_proj_ln_applies(qkv, heads, packs['out'], prec, lens.order):
            # attention of both heads + out-projection + dropout + residual + LayerNorm in ONE launch (same bits as the two launches below)
            att, lse, z1, y1, mean1, rstd1, *rest = ops.attn_proj_ln_fwd(qkv, lens.i32, heads, s_attn, p_attn, packs['out'], out_b, x, ln1_w, ln1_b, None,
                                                                         seed_pre=s_ln1, p_pre=p_attn, shadow=sh, seed_offset=so, prec=prec)
            y1g = rest[0] if sh else y1
        else:
            att, lse = ops.attention_fwd(qkv, lens.i32, heads, s_attn, p_attn, prec=prec, seed_offset=so, ctx_dtype=hd, order=lens.order)   # 16-bit modes: 16-bit context
            if ops.proj_ln_applies(att, packs['out'], prec):   # out-projection + dropout + residual + LayerNorm: one launch, z1 straight from LDS
                z1, y1, mean1, rstd1, *rest = ops.proj_ln_fwd(att, packs['out'], out_b, x, ln1_w, ln1_b, None, lens.i32, seed_pre=s_ln1, p_pre=p_attn,
                                                             shadow=sh, seed_offset=so, prec=prec)
                y1g = rest[0] if sh else y1
            else:
                z1 = ops.conv_gemm(att, packs['out'], out_b, lens=L, halo=0, prec=prec)
                ln1 = ops.ln_fwd(z1, x, ln1_w, ln1_b, None, lens.i32, seed_pre=s_ln1, p_pre=p_attn, shadow=sh, seed_offset=so, prec=prec)
                y1, mean1, rstd1 = ln1[:3]
                y1g = ln1[3] if sh else y1             # the copy the GEMMs read
        fused = ops.ff_pair_applies(y1g, packs['c1'], packs['c2'], prec)
        y2 = qkv_next = hmask = None
        if fused and ops._FF_LN:   # ... and the block's second LayerNorm on the output tile while it is still in LDS
            if next_in is not None and not ops.next_qkv_applies(next_in[0], prec):
                next_in = None
            need_bwd = any(ctx.needs_input_grad)
            z2, h, y2, mean2, rstd2, *rest = ops.ff_pair_ln(y1g, packs['c1'], packs['c2'], c1_b, c2_b, L, y1, ln2_w, ln2_b, film, seed_pre=s_ln2,
                                                            p_pre=p_conv, seed_offset=so, prec=prec, rows_exist=lens.exist, next_in=next_in,
                                                            need_h=need_bwd,       # inference: the 2 KB/token hidden tensor is not written
                                                            want_mask=need_bwd)    # training: + the ReLU sign words the backward reads instead of h
            hmask = rest.pop() if need_bwd else None
            qkv_next = rest[0] if rest else None         # ... and the next block's in-projection on the normalised tile
        elif fused:    # conv1 + ReLU + conv2 in ONE launch, the 1024-wide hidden tile consumed from LDS (h is still written: weight gradients)
            z2, h = ops.ff_pair(y1g, packs['c1'], packs['c2'], c1_b, c2_b, L, prec=prec, rows_exist=lens.exist)
        else:
            h = ops.conv_gemm(y1g, packs['c1'], c1_b, relu=True, lens=L, halo=1, out_dtype=hd, prec=prec, rows_exist=lens.exist)   # conv2 reads one row past the end
            z2 = ops.conv_gemm(h, packs['c2'], c2_b, lens=L, halo=0, prec=prec, rows_exist=lens.exist)
        if y2 is None:
            y2, mean2, rstd2 = ops.ln_fwd(z2, y1, ln2_w, ln2_b, film, lens.i32, seed_pre=s_ln2, p_pre=p_conv, seed_offset=so, prec=prec)
        ctx.save_for_backward(x, film, qkv, att, lse, z1, mean1, rstd1, y1g, h, z2, mean2, rstd2, ln1_w, ln1_b, ln2_w, ln2_b)
        ctx.lens, ctx.packs, ctx.heads = lens, packs, heads
        ctx.drop = (p_attn, p_conv, s_attn, s_ln1, s_ln2)
        ctx.prec, ctx.sink, ctx.fused, ctx.seed_offset = prec, rt.sink, fused, so
        ctx.hmask = hmask                          # (an int32 side tensor of the fused forward; None elsewhere)
        if qkv_next is not None:
            ctx.mark_non_differentiable(qkv_next)
        ctx.set_materialize_grads(False)           # else autograd hands backward a zero tensor the size of qkv_next (33 MB fill per block)
        return y2, qkv_next

    @staticmethod
    def backward(ctx, dy2, _dqkv_next=None):
        x, film, qkv, att, lse, z1, mean1, rstd1, y1, h, z2, mean2, rstd2, ln1_w, ln1_b, ln2_w, ln2_b = ctx.saved_tensors
        lens, packs = ctx.lens, ctx.packs
        if lens.exist is not None:
            raise NotImplementedError('Lengths.exist (padded-shape buckets / stand-alone rows) is a forward-only feature')
        p_attn, p_conv, s_attn, s_ln1, s_ln2 = ctx.drop
        dy2 = dy2.contiguous()
        L = lens.i32
        B, _, D = x.shape
        Fc = h.shape[2]
        pad = ops.ZeroArena.padded
        # the step's arena (trainer) or, outside a trainer step, one fill for this block's accumulators
        arena = packs['in'].rt.arena or ops.ZeroArena(x.device, 6 * pad(D) + pad(B * 2 * D) + 2 * pad(3 * D * Fc) + pad(Fc) + pad(D * D) + pad(3 * D * D) + pad(3 * D) + 64)
        P = packs.get('params', {})
        prec = ctx.prec
        sk = {k: _sink(v, ctx.sink) for k, v in P.items()}
        g = sk.get
        sh = ops.gemm_shadow(prec)
        so = ctx.seed_offset
        if ctx.fused and ops._FF_BLOCK_BWD and sh and (film is None or film.stride(-1) == 1):
            # LayerNorm2-backward (prologue) -> conv2^T -> ReLU mask -> conv1^T -> LayerNorm1-backward (epilogue): ONE launch
            # ... -> the out-projection's input gradient (datt) on the same tile
            fuse_datt = att.dtype == ops._H16[prec] and ops._FF_BLOCK_DATT
            dz1, dh, dproj, dff, dfilm, dln2_w, dln2_b, dln1_w, dln1_b, datt = ops.ff_block_bwd(
                dy2, z2, mean2, rstd2, ln2_w, ln2_b, film, packs['c1'], packs['c2'], L, h, z1, mean1, rstd1, ln1_w, ln1_b,
                seed2=s_ln2, p2=p_conv, seed1=s_ln1, p1=p_attn, seed_offset=so, prec=prec, arena=arena,
                sinks={'ln2_w': g('ln2_w'), 'ln2_b': g('ln2_b'), 'ln1_w': g('ln1_w'), 'ln1_b': g('ln1_b')},
                out_pack=packs['out'] if fuse_datt else None, hmask=ctx.hmask)
            dc2_w, dc2_b = ops.conv_wgrad(dff, h, packs['c2'], L, 0, arena=arena, w_sink=g('c2_w'), b_sink=g('c2_b'), prec=prec, defer=True)
        else:
            datt = None
            r2 = ops.ln_bwd(dy2, z2, mean2, rstd2, ln2_w, ln2_b, film, L, want_da=p_conv > 0, seed_pre=s_ln2, seed_offset=so, prec=prec,
                            p_pre=p_conv, arena=arena, w_sink=g('ln2_w'), b_sink=g('ln2_b'), shadow=sh)
            dz2, da2, dln2_w, dln2_b, dfilm = r2[:5]
            dff = r2[5] if sh else (da2 if da2 is not None else dz2)      # gradient w.r.t. the conv2 output, as the GEMMs read it
            # (before the pair below accumulates onto dz2 in place: without a 16-bit shadow and without dropout dff IS dz2)
            dc2_w, dc2_b = ops.conv_wgrad(dff, h, packs['c2'], L, 0, arena=arena, w_sink=g('c2_w'), b_sink=g('c2_b'), prec=prec, defer=True)
            if ctx.fused:  # conv2^T -> ReLU mask -> conv1^T in one launch, accumulated into the residual-branch gradient
                dy1, dh = ops.ff_pair(dff, packs['c1'], packs['c2'], None, None, L, backward=True, aux=h, out=dz2, accumulate=True, prec=prec)
            else:
                dh = ops.conv_gemm(dff, packs['c2'], None, transpose=True, relu_aux=h, lens=L, halo=1, out_dtype=h.dtype, prec=prec)
                dy1 = ops.conv_gemm(dh, packs['c1'], None, transpose=True, out=dz2, accumulate=True, lens=L, halo=0, prec=prec)  # + residual branch
            r1 = ops.ln_bwd(dy1, z1, mean1, rstd1, ln1_w, ln1_b, None, L, want_da=p_attn > 0, seed_pre=s_ln1, seed_offset=so, prec=prec,
                            p_pre=p_attn, arena=arena, w_sink=g('ln1_w'), b_sink=g('ln1_b'), shadow=sh)
            dz1, da1, dln1_w, dln1_b = r1[:4]
            dproj = r1[5] if sh else (da1 if da1 is not None else dz1)
        dc1_w, dc1_b = ops.conv_wgrad(dh, y1, packs['c1'], L, 1, arena=arena, w_sink=g('c1_w'), b_sink=g('c1_b'), prec=prec, defer=True)
        dout_w, dout_b = ops.conv_wgrad(dproj, att, packs['out'], L, 0, arena=arena, w_sink=g('out_w'), b_sink=g('out_b'), prec=prec, defer=True)
        if datt is None:
            datt = ops.conv_gemm(dproj, packs['out'], None, transpose=True, lens=L, halo=0, prec=prec, out_dtype=att.dtype)   # stored like the context
        dqkv = ops.attention_bwd(qkv, att, datt, lse, L, ctx.heads, s_attn, p_attn, out_dtype=qkv.dtype, prec=prec, seed_offset=so, order=lens.order)
        din_w, din_b = ops.conv_wgrad(dqkv, x, packs['in'], L, 0, arena=arena, w_sink=g('in_w'), b_sink=g('in_b'), prec=prec, defer=True)
        dx = ops.conv_gemm(dqkv, packs['in'], None, transpose=True, out=dz1, accumulate=True, lens=L, halo=0, prec=prec)  # + residual branch
        return (dx, dfilm, None, None, None, None,
                din_w, din_b, dout_w, dout_b, dln1_w, dln1_b, dc1_w, dc1_b, dc2_w, dc2_b, dln2_w, dln2_b, None, None)


# ----------------------------------------------------------------------------------------------------------------------
# backward cuts: a trainer (trainer.Trainer) runs the backward pass in PHASES so that the gradient buckets of a finished phase are
# exchanged while the next one computes.  ``cut(rt, level, t)`` ends a phase at tensor ``t``: the graph downstream sees a fresh leaf,
# and the trainer later calls ``t.backward(leaf.grad)``.  Levels are numbered in BACKWARD order (1 = first cut the backward meets).
# ----------------------------------------------------------------------------------------------------------------------
def cut(rt, level, t):
    sp = None if rt is None else rt.backward_split
    if sp is None or level > rt.cut_levels or not t.requires_grad:
        return t
    leaf = t.detach().requires_grad_(True)
    sp.append((level, t, leaf))
    return leaf


# ----------------------------------------------------------------------------------------------------------------------
# accent-encoder front end (model.py:687-706): prenet 3 x [conv k3 -> ReLU -> LN -> dropout], + energy/pitch/position, mask.
# Two Functions -- layer 0 (80 -> 1024) and layers 1, 2 + the embedding sum -- so that a trainer can cut the backward between them:
# the 1024 x 1024 layer's 12.6 MB gradient is then exchanged while layer 0's backward still runs (trainer.Trainer, phase D).
# ----------------------------------------------------------------------------------------------------------------------
class AccentFront0Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel, lens, packs, p_drop, training, c0_w, c0_b, l0_w, l0_b):
        p = p_drop if training else 0.0
        seed = next_seed() if training else 0
        x0 = ops.transpose(mel.contiguous())                                   # (B, T, n_mel) channels-last
        L = lens.i32
        rt = packs['p0'].rt
        prec = rt.precision
        hd = ops.hidden_dtype(prec)                                           # 1024-wide tensors: bf16 in bf16 operand mode
        so = rt.seed_offset
        h0 = ops.conv_gemm(x0, packs['p0'], c0_b, relu=True, lens=L, halo=2, out_dtype=hd, prec=prec, rows_exist=lens.exist)   # three stacked k=3 convs: halos 2, 1, 0
        y0, m0, r0 = ops.ln_fwd(h0, None, l0_w, l0_b, None, L, seed_post=seed, p_post=p, halo=2, seed_offset=so, prec=prec)
        ctx.save_for_backward(x0, h0, m0, r0, l0_w, l0_b)
        ctx.lens, ctx.packs, ctx.p, ctx.seed = lens, packs, p, seed
        ctx.prec, ctx.sink, ctx.seed_offset = prec, rt.sink, so
        return y0

    @staticmethod
    def backward(ctx, dy0):
        x0, h0, m0, r0, l0_w, l0_b = ctx.saved_tensors
        lens, packs, p = ctx.lens, ctx.packs, ctx.p
        arena = packs['p0'].rt.arena
        L = lens.i32
        sk = {k: _sink(v, ctx.sink) for k, v in packs.get('params', {}).items()}
        g = sk.get
        dz0, _, dl0_w, dl0_b, _ = ops.ln_bwd(dy0.contiguous(), h0, m0, r0, l0_w, l0_b, None, L, relu_mask=True, seed_post=ctx.seed, p_post=p,
                                              seed_offset=ctx.seed_offset, prec=ctx.prec, w_sink=g('l0_w'), b_sink=g('l0_b'), halo=2, arena=arena)
        dc0_w, dc0_b = ops.conv_wgrad(dz0, x0, packs['p0'], L, 2, w_sink=g('c0_w'), b_sink=g('c0_b'), prec=ctx.prec, arena=arena)
        return None, None, None, None, None, dc0_w, dc0_b, dl0_w, dl0_b


class AccentFront12Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, energy, pitch, lens, packs, pe, p_drop, training,
                c1_w, c1_b, l1_w, l1_b, c2_w, c2_b, l2_w, l2_b, we, be, wp, bp):
        p = p_drop if training else 0.0
        seeds = [next_seed() if training else 0 for _ in range(2)]
        L = lens.i32
        rt = packs['p1'].rt
        prec = rt.precision
        hd = ops.hidden_dtype(prec)
        so = rt.seed_offset
        h1 = ops.conv_gemm(y0, packs['p1'], c1_b, relu=True, lens=L, halo=1, out_dtype=hd, prec=prec, rows_exist=lens.exist)
        y1, m1, r1 = ops.ln_fwd(h1, None, l1_w, l1_b, None, L, seed_post=seeds[0], p_post=p, halo=1, seed_offset=so, prec=prec)
        h2 = ops.conv_gemm(y1, packs['p2'], c2_b, relu=True, lens=L, halo=0, prec=prec, rows_exist=lens.exist)
        y2, m2, r2 = ops.ln_fwd(h2, None, l2_w, l2_b, None, L, seed_post=seeds[1], p_post=p, halo=0, seed_offset=so, prec=prec)
        energy, pitch = energy.contiguous(), pitch.contiguous()
        out = ops.accent_sum(y2, energy, pitch, we, be, wp, bp, pe, lens.i32)
        ctx.save_for_backward(y0, h1, m1, r1, y1, h2, m2, r2, energy, pitch, l1_w, l1_b, l2_w, l2_b)
        ctx.lens, ctx.packs, ctx.p, ctx.seeds = lens, packs, p, seeds
        ctx.emb_params = (we, be, wp, bp)
        ctx.prec, ctx.sink, ctx.seed_offset = prec, rt.sink, so
        return out

    @staticmethod
    def backward(ctx, dout):
        y0, h1, m1, r1, y1, h2, m2, r2, energy, pitch, l1_w, l1_b, l2_w, l2_b = ctx.saved_tensors
        lens, packs, p, seeds = ctx.lens, ctx.packs, ctx.p, ctx.seeds
        so = ctx.seed_offset
        arena = packs['p1'].rt.arena
        dout = ops.mask_rows(dout.contiguous(), lens.i32)
        dwe, dbe, dwp, dbp = ops.scalar_conv_wgrad(dout, energy, pitch, lens.i32, sinks=tuple(_sink(q, ctx.sink) for q in ctx.emb_params))
        L = lens.i32
        prec = ctx.prec
        sk = {k: _sink(v, ctx.sink) for k, v in packs.get('params', {}).items()}
        g = sk.get
        dz2, _, dl2_w, dl2_b, _ = ops.ln_bwd(dout, h2, m2, r2, l2_w, l2_b, None, L, relu_mask=True, seed_post=seeds[1], p_post=p, seed_offset=so, prec=prec,
                                              w_sink=g('l2_w'), b_sink=g('l2_b'), halo=0, arena=arena)
        dc2_w, dc2_b = ops.conv_wgrad(dz2, y1, packs['p2'], L, 0, w_sink=g('c2_w'), b_sink=g('c2_b'), prec=prec, arena=arena)
        dy1 = ops.conv_gemm(dz2, packs['p2'], None, transpose=True, lens=L, halo=1, out_dtype=h1.dtype, prec=prec)
        dz1, _, dl1_w, dl1_b, _ = ops.ln_bwd(dy1, h1, m1, r1, l1_w, l1_b, None, L, relu_mask=True, seed_post=seeds[0], p_post=p, seed_offset=so, prec=prec,
                                              w_sink=g('l1_w'), b_sink=g('l1_b'), halo=1, arena=arena)
        dc1_w, dc1_b = ops.conv_wgrad(dz1, y0, packs['p1'], L, 1, w_sink=g('c1_w'), b_sink=g('c1_b'), prec=prec, arena=arena)
        dy0 = ops.conv_gemm(dz1, packs['p1'], None, transpose=True, lens=L, halo=2, out_dtype=y0.dtype, prec=prec)
        return (dy0, None, None, None, None, None, None, None,
                dc1_w, dc1_b, dl1_w, dl1_b, dc2_w, dc2_b, dl2_w, dl2_b, dwe, dbe, dwp, dbp)


class SplitFilmFn(torch.autograd.Function):
    """(B, nb, 2C) FiLM parameters -> nb contiguous (B, 2C) tensors, one per FFT block.  Slicing ``film[:, i, :]`` per block costs
    the backward a zero fill + copy + add per slice (SelectBackward: ~22 tiny launches per step); here it is one copy forward and
    one stack backward."""

    @staticmethod
    def forward(ctx, film):
        ctx.shape = film.shape
        t = film.transpose(0, 1).contiguous()
        return tuple(t[i] for i in range(film.shape[1]))

    @staticmethod
    def backward(ctx, *grads):
        B, nb, C2 = ctx.shape
        ref = next(g for g in grads if g is not None)
        return torch.stack([g if g is not None else torch.zeros_like(ref) for g in grads], dim=1)


class FilmAffineFn(torch.autograd.Function):
    """StyleAdapter's tail (model.py:779-800) for all FFT blocks at once: scalar post-multiplier affine on the predicted gammas / betas,
    (gamma | beta) concatenation and the per-block split -- one launch forward, one backward (was ~5 + ~10 ATen launches incl. two
    full reductions for the post-multiplier gradients).  Returns nb contiguous (B, 2C) tensors (views of one block-major buffer)."""

    @staticmethod
    def forward(ctx, gammas, betas, pm, nb, rt):
        gammas, betas = gammas.contiguous(), betas.contiguous()
        pmc = None if pm is None else pm.detach().contiguous()
        film = ops.film_affine_fwd(gammas, betas, pmc, nb)
        ctx.save_for_backward(gammas, betas, pmc)
        ctx.nb, ctx.rt, ctx.has_pm = nb, rt, pm is not None
        whole = film.detach()                      # (nb, B, 2C): for the (B, nb, 2C) tensors of the reference's outputs[1] (strided views, no copy)
        ctx.mark_non_differentiable(whole)
        ctx.set_materialize_grads(False)           # blocks without a gradient arrive as None (film_affine_bwd skips them), not as zero fills
        return tuple(film[i] for i in range(nb)) + (whole,)

    @staticmethod
    def backward(ctx, *grads):
        gammas, betas, pm = ctx.saved_tensors
        dg, db, dpm = ops.film_affine_bwd(list(grads[:ctx.nb]), gammas, betas, pm, arena=None if ctx.rt is None else ctx.rt.arena)
        return dg, db, dpm, None, None


class MeanPoolFn(torch.autograd.Function):
    """sum over time / length (model.py:714)."""

    @staticmethod
    def forward(ctx, x, lens, rt=None):
        ctx.lens, ctx.N = lens, x.shape[1]
        return ops.mean_pool(x, lens.i32, arena=None if rt is None else rt.arena)

    @staticmethod
    def backward(ctx, dout):
        return ops.mean_pool_bwd(dout.contiguous(), ctx.lens.i32, ctx.N), None, None


class EmbedPosFn(torch.autograd.Function):
    """mask(embedding[symbols] + position) (model.py:597-604)."""

    @staticmethod
    def forward(ctx, symbols, emb, pe, lens, rt=None):
        symbols = symbols.contiguous()
        ctx.save_for_backward(symbols)
        ctx.lens, ctx.rows, ctx.rt = lens, emb.shape[0], rt
        ctx.emb, ctx.sink = emb, bool(rt is not None and rt.sink)
        return ops.add_pos(None, symbols, emb, pe, lens.i32)

    @staticmethod
    def backward(ctx, dout):
        (symbols,) = ctx.saved_tensors
        arena = None if ctx.rt is None else ctx.rt.arena
        return None, ops.embedding_bwd(dout.contiguous(), symbols, ctx.lens.i32, ctx.rows, arena=arena, sink=_sink(ctx.emb, ctx.sink)), None, None, None


class AddPosFn(torch.autograd.Function):
    """mask(x + position) (model.py:554-557)."""

    @staticmethod
    def forward(ctx, x, pe, lens):
        ctx.lens = lens
        return ops.add_pos(x.contiguous(), None, None, pe, lens.i32)

    @staticmethod
    def backward(ctx, dout):
        return ops.mask_rows(dout.contiguous(), ctx.lens.i32), None, None


# ----------------------------------------------------------------------------------------------------------------------
# Gaussian upsampling (model.py:417-510)
# ----------------------------------------------------------------------------------------------------------------------
class GaussianUpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, dur_float, dur_int, energy, pitch, lens, n_frames, rt, wd, bd, we, be, wp, bp, wr, br):
        dur_float, energy, pitch = dur_float.contiguous(), energy.contiguous(), pitch.contiguous()
        xs, z, sigma = ops.upsample_prep(enc.contiguous(), dur_float, energy, pitch, wd, bd, we, be, wp, bp, wr, br, lens.i32)
        mu, _totals = ops.duration_scan(dur_int.contiguous())
        xup, weights = ops.upsample_fwd(xs, mu, sigma, lens.i32, n_frames)
        ctx.save_for_backward(xs, z, sigma, mu, weights, dur_float, energy, pitch, wd, bd, wr)
        ctx.lens = lens
        ctx.params = (wd, bd, we, be, wp, bp, wr, br)
        ctx.sink = bool(rt is not None and rt.sink)
        ctx.rt = rt
        ctx.mark_non_differentiable(weights)
        ctx.set_materialize_grads(False)           # no 20 MB zero tensor for the alignment's (unused) gradient slot
        return xup, weights

    @staticmethod
    def backward(ctx, dxup, _dweights):
        xs, z, sigma, mu, weights, dur_float, energy, pitch, wd, bd, wr = ctx.saved_tensors
        lens = ctx.lens
        dxs, dsigma = ops.upsample_bwd(dxup.contiguous(), xs, mu, sigma, weights, lens.i32, arena=None if ctx.rt is None else ctx.rt.arena)
        pwd, pbd, pwe, pbe, pwp, pbp, pwr, pbr = ctx.params
        sk = lambda q: _sink(q, ctx.sink)
        swr = sk(pwr)
        dxs, dz, dwr, dbr = ops.upsample_sym_bwd(dxs, dsigma, xs, z, dur_float, lens.i32, wd, bd, wr,
                                                 dwr_sink=None if swr is None else swr.view(-1), dbr_sink=sk(pbr))
        dwe, dbe, dwp, dbp = ops.scalar_conv_wgrad(dxs, energy, pitch, lens.i32, sinks=(sk(pwe), sk(pbe), sk(pwp), sk(pbp)))
        dwd, dbd, _, _ = ops.scalar_conv_wgrad(wr.reshape(-1), dur_float, None, lens.i32, rowscale=dz, sinks=(sk(pwd), sk(pbd), None, None))
        denc = ops.mask_rows(dxs, lens.i32)  # the encoder output is zero-masked; rows >= len carry no gradient upstream
        return (denc, None, None, None, None, None, None, None, dwd, dbd, dwe, dbe, dwp, dbp,
                None if dwr is None else dwr.view(1, -1), dbr)


class MelProjectionFn(torch.autograd.Function):
    """Linear(D -> n_mel) + mask + transpose to (B, n_mel, T) (model.py:561-563)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pack, lens):
        ctx.prec = pack.rt.precision
        mel_cl = ops.conv_gemm(x, pack, bias, lens=lens.i32, mask_rows=True, halo=0, prec=ctx.prec)
        ctx.save_for_backward(x)
        ctx.pack, ctx.lens = pack, lens
        ctx.params, ctx.sink = (weight, bias), bool(pack.rt.sink)
        return ops.transpose(mel_cl)

    @staticmethod
    def backward(ctx, dmel):
        (x,) = ctx.saved_tensors
        d_cl = ops.mask_rows(ops.transpose(dmel.contiguous()), ctx.lens.i32)
        dw, db = ops.conv_wgrad(d_cl, x, ctx.pack, ctx.lens.i32, 0, arena=ctx.pack.rt.arena, w_sink=_sink(ctx.params[0], ctx.sink),
                                b_sink=_sink(ctx.params[1], ctx.sink), prec=ctx.prec)
        dx = ops.conv_gemm(d_cl, ctx.pack, None, transpose=True, lens=ctx.lens.i32, halo=0, prec=ctx.prec)
        return dx, dw, db, None, None
