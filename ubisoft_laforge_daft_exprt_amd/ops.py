"""Tensor-level wrappers over the C ABI (include/daft_exprt_hip.h).

PyTorch is used for device memory (caching allocator), the current HIP stream and autograd bookkeeping only; all
arithmetic on the path is done by libdaft_exprt_hip.so.  Every wrapper launches on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import lib

PRECISIONS = ('f32', 'bf16', 'fp16')


class Runtime:
    """Execution state of ONE model (or one loss object): operand precision, the pack epoch its optimiser bumps, its gradient
    sink and its launch log.  Nothing on the product path reads process-global mutable state: autograd Functions capture
    ``runtime.precision`` / ``runtime.sink`` in ``ctx`` at forward time and use the captured values in backward, so two models
    with different settings (or a precision flipped between a forward and its backward) cannot mix modes; backward runs on
    autograd engine threads and only touches the ``ctx`` it was given (threading row)."""

    def __init__(self, precision='f32'):
        self.precision = _check_precision(precision)
        self.pack_epoch = 0          # "an optimiser wrote the parameters through raw pointers" (optim.FusedAdam)
        self.sink = False            # backward kernels accumulate straight into pre-zeroed ``param.grad`` (ddp.GradientReducer)
        self.launch_log = None       # bench.py: (kind, ...) records of every launch, to price algorithmic FLOPs / bytes
        self.arena = None            # ZeroArena of the running training step (trainer.Trainer): one zero fill for all small accumulators
        self.arena_hint = 1 << 20    # floats; tracks the high-water mark of the previous steps
        self.seed_offset = None      # device int64 scalar added to every dropout seed on the device (graph replays bump it)
        self.defer_wgrad = False     # trainer.Trainer: the FFT blocks' weight gradients are queued and launched 8 layers at a time
        self.wgrad_queue = {}        # (taps, dy 16-bit, x 16-bit, precision) -> [(job fields, tensors kept alive)]; see flush_wgrads
        self.backward_split = None   # a list while a trainer runs the backward in phases: (level, tensor, leaf) per cut (functional.cut)
        self.cut_levels = 0          # cuts the trainer wants: 0 = none, 1 = at the accent embedding, 2 = + before the prenet, 3 = + inside the prenet
        self._packs = []             # weak refs to this runtime's PackedWeight objects (one-launch batched re-pack)
        self._tables = {}

    def set_precision(self, name):
        self.precision = _check_precision(name)

    def __deepcopy__(self, memo):
        new = Runtime(self.precision)          # a copied model gets its own state; packs re-register as they are copied
        new.pack_epoch = self.pack_epoch
        memo[id(self)] = new
        return new

    def invalidate_packs(self):
        self.pack_epoch += 1

    def record_launches(self, enable: bool):
        log = self.launch_log
        self.launch_log = [] if enable else None
        return log


def _check_precision(name):
    if name not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {name!r}")
    return name


# Default runtime: the configuration that kernel-level calls without a model use (tests, micro-benchmarks) and whose
# precision newly constructed models / losses START from.  A model never reads it after construction.
DEFAULT = Runtime('f32')


def set_precision(name: str) -> None:
    """'f32': exact-f32 MFMA operands (parity mode).  'bf16' / 'fp16': 16-bit MFMA operands, fp32 accumulate (throughput modes).
    Sets the default for models / losses constructed afterwards (and for kernel-level calls without a model); a live model is
    switched with ``model.set_precision(name)``."""
    DEFAULT.set_precision(name)


def get_precision() -> str:
    return DEFAULT.precision


def invalidate_packs(rt=None) -> None:
    (rt or DEFAULT).invalidate_packs()


def record_launches(enable: bool, rt=None):
    return (rt or DEFAULT).record_launches(enable)


def _half(prec) -> int:
    """operand-mode flag of the C ABI: 0 = exact f32, 1 = 16-bit operands (bf16 build, or fp16 in the ``_f16`` twins)"""
    return 1 if prec in ('bf16', 'fp16') else 0


_H16 = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def _fn(name, prec):
    """The C entry point serving operand precision ``prec``: the ``_f16`` twin in fp16 mode (include/daft_exprt_hip.h)."""
    return getattr(lib(), name + '_f16' if prec == 'fp16' else name)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name):
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError(f'{name} must be a float32 (or, in a 16-bit operand mode, bfloat16 / float16) tensor on the GPU (got {t.dtype} on {t.device})')
    return t


def _is_bf16(t):
    """1 when the tensor is stored in the 16-bit type of its operand mode (the C ABI's ``*_bf16`` storage flags)"""
    return int(t is not None and t.dtype in (torch.bfloat16, torch.float16))


def _check_h16(prec, *tensors):
    want = _H16.get(prec)
    for t in tensors:
        if t is not None and t.dtype in (torch.bfloat16, torch.float16) and t.dtype != want:
            raise TypeError(f'{t.dtype} tensor in {prec!r} operand mode')


def hidden_dtype(prec=None):
    """Storage type of the 1024-wide hidden activations: bf16 in bf16 operand mode (their HBM traffic bounds the step)."""
    return _H16.get(prec or DEFAULT.precision, torch.float32)


def _rows(t):
    """(B, N, C)-shaped tensor whose last dim is contiguous -> (ld, C); rows must be uniformly strided."""
    if t.stride(-1) != 1:
        raise ValueError('last dimension must be contiguous')
    ld = t.stride(-2)
    if t.dim() == 3 and t.stride(0) != t.shape[1] * ld:
        raise ValueError('batch rows must be densely stacked')
    return ld


def repack_all(rt=None, packs=None) -> int:
    """Re-packs every stale PackedWeight of the runtime (at its current precision) with ONE kernel launch - what follows an
    optimiser step; returns the number packed."""
    import struct
    rt = rt or DEFAULT
    prec = rt.precision
    live = [r() for r in rt._packs] if packs is None else list(packs)
    stale = []
    for pk in live:
        if pk is None or not pk.weight.is_cuda:
            continue
        img = pk._image(prec)
        if pk._current_key() != img.key:
            stale.append((pk, img))
    if not stale:
        return 0
    sig = (prec,) + tuple((pk.weight.data_ptr(), img.fwd.data_ptr(), img.bwd.data_ptr()) for pk, img in stale)
    table = rt._tables.get(sig)
    if table is None:                      # the descriptor records only change when buffers are (re)allocated: build them once (host memory)
        recs = b''.join(struct.pack('<QQQ8i', pk.weight.data_ptr(), img.fwd.data_ptr(), img.bwd.data_ptr(), pk.cout, pk.cin, pk.taps,
                                     img.dims[0], img.dims[1], img.dims[2], img.dims[3], 0) for pk, img in stale)
        table = ctypes.create_string_buffer(recs, len(recs))
        if len(rt._tables) > 8:
            rt._tables.clear()
        rt._tables[sig] = table
    _fn('dx_pack_weights_host', prec)(ctypes.addressof(table), len(stale), _half(prec), _stream())     # descriptors travel as kernel arguments
    for pk, img in stale:
        img.key = pk._current_key()
    return len(stale)


class _PackImage:
    """The MFMA-ready copies of one parameter at one operand precision."""
    __slots__ = ('fwd', 'bwd', 'dims', 'key', 'half')

    def __init__(self):
        self.fwd = self.bwd = self.dims = self.key = None
        self.half = 0


class PackedWeight:
    """MFMA-ready copies of one (Cout, Cin[, taps]) parameter, one image per operand precision in use, each refreshed when the
    parameter changes.  Belongs to a Runtime (``rt``): its precision is the default for calls that do not name one, and its
    pack epoch tells the images that an optimiser rewrote the parameter through raw pointers."""

    def __init__(self, weight: torch.Tensor, rt: Runtime = None):
        import weakref
        self.weight = weight
        self.rt = rt or DEFAULT
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.taps = weight.shape[2] if weight.dim() == 3 else 1
        self._images = {}
        self.rt._packs.append(weakref.ref(self))
        if len(self.rt._packs) > 4096:
            self.rt._packs[:] = [r for r in self.rt._packs if r() is not None]

    def __deepcopy__(self, memo):
        import copy
        new = PackedWeight(copy.deepcopy(self.weight, memo), copy.deepcopy(self.rt, memo))
        memo[id(self)] = new
        return new

    def _current_key(self):
        w = self.weight
        # the pack epoch stands for "an optimiser step happened": frozen weights (the pitch predictor of the loss) are not part of it
        return (w._version, w.data_ptr(), self.rt.pack_epoch if w.requires_grad else 0)

    def _image(self, prec) -> _PackImage:
        img = self._images.get(prec)
        w = self.weight
        if img is None or img.fwd.device != w.device:
            img = _PackImage()
            img.half = _half(prec)
            dt = _H16.get(prec, torch.float32)
            dims = (ctypes.c_int * 4)()
            lib().dx_pack_dims(self.cout, self.cin, img.half, ctypes.cast(dims, ctypes.c_void_p))
            img.dims = [int(d) for d in dims]
            img.fwd = torch.empty(self.taps * img.dims[0] * img.dims[1], dtype=dt, device=w.device)
            img.bwd = torch.empty(self.taps * img.dims[2] * img.dims[3], dtype=dt, device=w.device)
            self._images[prec] = img
        return img

    def image(self, prec=None) -> _PackImage:
        """The up-to-date image at ``prec`` (default: the runtime's precision)."""
        prec = prec or self.rt.precision
        img = self._image(prec)
        key = self._current_key()
        if key != img.key:                     # (the padded speaker-logit layer's copy bumps its _version under capture: recorded too)
            _fn('dx_pack_weights', prec)(_p(self.weight.detach()), _p(img.fwd), _p(img.bwd), self.cout, self.cin, self.taps, img.half, _stream())
            img.key = key
        return img

    def refresh(self, prec=None):
        self.image(prec)
        return self


def _log(pack_or_rt, rec):
    rt = pack_or_rt.rt if isinstance(pack_or_rt, PackedWeight) else (pack_or_rt or DEFAULT)
    if rt.launch_log is not None:
        rt.launch_log.append(rec)


def conv_gemm(x, pack: PackedWeight, bias=None, *, transpose=False, relu=False, post_scale=None, post_shift=None,
              relu_aux=None, out=None, accumulate=False, lens=None, mask_rows=False, out_scale=1.0, B=None, N=None, halo=-1,
              out_dtype=torch.float32, prec=None, rows_exist=None):
    """y = epilogue(conv(x)).  ``transpose=True`` runs the input-gradient convolution (x is dY, result is dX).
    ``prec``: operand precision (default: the pack's runtime); backward passes hand in the precision their forward captured."""
    _chk(x, 'x')
    prec = prec or pack.rt.precision
    _check_h16(prec, x, out, relu_aux)
    img = pack.image(prec)
    if x.dim() == 2:
        B_, N_ = (1, x.shape[0]) if B is None else (B, N)
    else:
        B_, N_ = x.shape[0], x.shape[1]
    cin, cout = (pack.cout, pack.cin) if transpose else (pack.cin, pack.cout)
    if x.shape[-1] != cin:
        raise ValueError(f'expected {cin} input channels, got {x.shape[-1]}')
    ldx = _rows(x)
    if out is None:
        out = torch.empty(*x.shape[:-1], cout, dtype=out_dtype, device=x.device)
    ldy = _rows(out)
    _log(pack, ('conv', B_ * N_, N_, cin, cout, pack.taps))
    _fn('dx_conv_gemm', prec)(_p(x), ldx, _p(img.bwd if transpose else img.fwd), _p(bias), _p(out), ldy, B_, N_, cin, cout, pack.taps,
                       img.half, int(relu), _p(post_scale), _p(post_shift), _p(relu_aux),
                       0 if relu_aux is None else _rows(relu_aux), int(accumulate), _p(lens), int(mask_rows), float(out_scale), int(halo),
                       _is_bf16(x), _is_bf16(out), _is_bf16(relu_aux), _p(rows_exist), _stream())
    return out


_FF_FUSED_MIN_TILES = int(os.environ.get('DX_FF_FUSED_MIN_TILES', '96'))       # 126-token tiles (the frame axis)
_FF_FUSED_MIN_TILES62 = int(os.environ.get('DX_FF_FUSED_MIN_TILES62', '64'))   # 62-token tiles (short batches: the symbol axis)


def ff_pair_applies(x, pack1: PackedWeight, pack2: PackedWeight, prec) -> bool:
    """The fused feed-forward kernel serves bf16 operand mode at the reference shape (128 -> F -> 128, k = 3) when the launch has
    enough 126-token tiles to occupy the chip: one workgroup per tile runs ~40 us whatever the grid, so the short symbol-level
    batches (48 tiles at C2: measured 40 vs 37 us forward, 55 vs 42 us backward) stay on the two-launch path."""
    return (prec in _H16 and x.dtype == _H16[prec] and x.dim() == 3 and x.shape[2] == 128 and pack1.taps == 3 and pack2.taps == 3
            and pack1.cin == 128 and pack2.cout == 128 and pack2.cin == pack1.cout and pack1.cout % 128 == 0
            and (x.shape[0] * ((x.shape[1] + 125) // 126) >= _FF_FUSED_MIN_TILES or x.shape[0] * ((x.shape[1] + 61) // 62) >= _FF_FUSED_MIN_TILES62))


def ff_pair(x, pack1: PackedWeight, pack2: PackedWeight, bias1, bias2, lens, *, backward=False, aux=None, out=None, accumulate=False,
            halo=1, prec=None, rows_exist=None):
    """The conv feed-forward pair in ONE launch (bf16 operand mode; csrc/dx_ffpair.hip).
    forward : (z, h)  with h = relu(conv1(x) + b1) [bf16, kept for the weight gradients], z = conv2(h) + b2 [fp32]
    backward: (dx, dh) with x = d(loss)/dz as bf16, dh = conv2^T(x) masked by ``aux`` = h > 0 [bf16], dx (+)= conv1^T(dh)
    ``pack1`` / ``pack2`` are the packs of conv1 (F, 128, 3) and conv2 (128, F, 3) in both directions."""
    prec = pack1.rt.precision if prec is None else prec
    if prec not in _H16 or x.dtype != _H16[prec]:
        raise TypeError('ff_pair runs in a 16-bit operand mode on an activation tensor of that type')
    B, N, D = x.shape
    Fc = pack1.cout
    if D != 128 or pack1.cin != 128 or pack2.cout != 128 or pack2.cin != Fc or pack1.taps != 3 or pack2.taps != 3:
        raise ValueError('ff_pair is built for conv(k=3, 128 -> F) -> conv(k=3, F -> 128)')
    i1, i2 = pack1.image(prec), pack2.image(prec)
    wa, wb = (i2.bwd, i1.bwd) if backward else (i1.fwd, i2.fwd)
    h = torch.empty(B, N, Fc, dtype=_H16[prec], device=x.device)
    if out is None:
        out = torch.empty(B, N, 128, dtype=torch.float32, device=x.device)
    _log(pack1, ('ffpair', B * N, N, 128, Fc, 3))
    _fn('dx_ff_pair', prec)(_p(x), _rows(x), _p(wa), _p(wb), _p(bias1), _p(bias2), _p(aux), 0 if aux is None else _rows(aux), _p(h), _rows(h),
                     _p(out), _rows(out), B, N, Fc, int(not backward), int(accumulate), _p(lens), int(halo), _p(rows_exist), _stream())
    return out, h


_FF_LN = os.environ.get('DX_FF_LN', '1') != '0'


_FF_QKV = os.environ.get('DX_FF_QKV', '1') != '0'


def next_qkv_applies(pack_in: PackedWeight, prec) -> bool:
    """The next block's attention in-projection can ride in the ff_pair epilogue: 128 -> 384, one tap, a 16-bit operand mode."""
    return _FF_QKV and _half(prec) == 1 and pack_in.taps == 1 and pack_in.cin == 128 and pack_in.cout == 384


_FF_MASK = os.environ.get('DX_FF_MASK', '1') != '0'


def ff_pair_ln(x, pack1: PackedWeight, pack2: PackedWeight, bias1, bias2, lens, res, ln_w, ln_b, film, *, seed_pre=0, p_pre=0.0, seed_offset=None,
               halo=1, prec=None, rows_exist=None, next_in=None, need_h=True, want_mask=False):
    """The forward pair with the block's second LayerNorm folded into its epilogue (dx_ff_pair_ln).  Returns (z, h, y, mean, rstd):
    z = res + dropout(pair output) [fp32, what ln_fwd leaves in its input], y = mask(FiLM(LN(z))).
    ``next_in`` = (PackedWeight, bias) of the NEXT block's attention in-projection: the launch also produces that block's qkv
    (dx_ff_pair_ln_qkv) and a sixth value, the 16-bit (B, N, 384) tensor, is returned.
    ``need_h=False`` (forward-only calls): the hidden tensor is not written and ``h`` is returned as None.
    ``want_mask``: one more value is appended to the result, the ReLU sign words of the hidden activation (int32 tensor, the C ABI's
    ``hmask``) that ``ff_block_bwd`` takes instead of re-reading ``h``."""
    prec = pack1.rt.precision if prec is None else prec
    B, N, D = x.shape
    Fc = pack1.cout
    i1, i2 = pack1.image(prec), pack2.image(prec)
    h = torch.empty(B, N, Fc, dtype=_H16[prec], device=x.device) if need_h else None
    z = torch.empty(B, N, 128, dtype=torch.float32, device=x.device)
    y = torch.empty_like(z)
    mean = torch.empty(B, N, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, N, dtype=torch.float32, device=x.device)
    mask = torch.empty(B * ((N + 61) // 62), Fc // 128, 4, 2, 64, dtype=torch.int32, device=x.device) if (want_mask and _FF_MASK) else None     # (room for either tile width)
    tail = (mask,) if want_mask else ()
    _log(pack1, ('ffpair', B * N, N, 128, Fc, 3))
    if next_in is not None:
        pq, bq = next_in
        qkv = torch.empty(B, N, 384, dtype=_H16[prec], device=x.device)
        _log(pq, ('conv', B * N, N, 128, 384, 1))
        _fn('dx_ff_pair_ln_qkv', prec)(_p(x), _rows(x), _p(i1.fwd), _p(i2.fwd), _p(bias1), _p(bias2), _p(h), Fc if h is None else _rows(h), _p(z), B, N, Fc, _p(lens), int(halo),
                                       _p(rows_exist), _p(res), _p(ln_w), _p(ln_b), _p(film), 0 if film is None else film.stride(0), _p(y), _p(mean),
                                       _p(rstd), seed_pre, float(p_pre), _p(seed_offset), _p(pq.image(prec).fwd), _p(bq), _p(qkv), _p(mask), _stream())
        return (z, h, y, mean, rstd, qkv) + tail
    _fn('dx_ff_pair_ln', prec)(_p(x), _rows(x), _p(i1.fwd), _p(i2.fwd), _p(bias1), _p(bias2), _p(h), Fc if h is None else _rows(h), _p(z), B, N, Fc, _p(lens), int(halo),
                               _p(rows_exist), _p(res), _p(ln_w), _p(ln_b), _p(film), 0 if film is None else film.stride(0), _p(y), _p(mean), _p(rstd),
                               seed_pre, float(p_pre), _p(seed_offset), _p(mask), _stream())
    return (z, h, y, mean, rstd) + tail




def ff_pair_lnbwd(x, pack1: PackedWeight, pack2: PackedWeight, lens, aux, out, z, mean, rstd, ln_w, ln_b, *, seed_pre=0, p_pre=0.0, seed_offset=None,
                  halo=1, prec=None, arena=None, w_sink=None, b_sink=None):
    """The input-gradient pair with the backward of the block's first LayerNorm in its epilogue (dx_ff_pair_lnbwd).  ``out`` holds the
    residual-branch gradient on entry and dz1 on return.  Returns (dz1 (= out), dh, dg_16bit, dw or None, db or None)."""
    prec = pack1.rt.precision if prec is None else prec
    B, N, D = x.shape
    Fc = pack1.cout
    i1, i2 = pack1.image(prec), pack2.image(prec)
    dh = torch.empty(B, N, Fc, dtype=_H16[prec], device=x.device)
    dg = torch.empty(B, N, 128, dtype=_H16[prec], device=x.device)
    dw = w_sink if w_sink is not None else _zeros(arena, 128, device=x.device)
    db = b_sink if b_sink is not None else _zeros(arena, 128, device=x.device)
    _log(pack1, ('ffpair', B * N, N, 128, Fc, 3))
    _fn('dx_ff_pair_lnbwd', prec)(_p(x), _rows(x), _p(i2.bwd), _p(i1.bwd), _p(aux), _rows(aux), _p(dh), _rows(dh), _p(out), B, N, Fc, _p(lens), int(halo),
                                  _p(z), _p(mean), _p(rstd), _p(ln_w), _p(ln_b), _p(dg), _p(dw), _p(db), seed_pre, float(p_pre), _p(seed_offset), _stream())
    return out, dh, dg, (None if w_sink is not None else dw), (None if b_sink is not None else db)


_FF_BLOCK_BWD = os.environ.get('DX_FF_BLOCK_BWD', '1') != '0'
_FF_BLOCK_DATT = os.environ.get('DX_FF_BLOCK_DATT', '1') != '0'


def ff_block_bwd(dy2, z2, mean2, rstd2, ln2_w, ln2_b, film, pack1: PackedWeight, pack2: PackedWeight, lens, aux, z1, mean1, rstd1, ln1_w, ln1_b, *,
                 seed2=0, p2=0.0, seed1=0, p1=0.0, seed_offset=None, halo=1, prec=None, arena=None, sinks=None, out_pack=None, hmask=None):
    """LayerNorm2-backward -> input-gradient pair -> LayerNorm1-backward in one launch (dx_ff_block_bwd).  ``sinks``: dict with optional
    pre-zeroed ``.grad`` tensors 'ln2_w', 'ln2_b', 'ln1_w', 'ln1_b'.  ``out_pack``: the attention out-projection's PackedWeight (128 x 128):
    the launch then also produces datt = dg1 x W_out.  Returns (dz1, dh, dg1_16bit, dg2_16bit, dfilm or None, dln2_w, dln2_b, dln1_w,
    dln1_b, datt or None) -- the four affine gradients are None where a sink took them."""
    prec = pack1.rt.precision if prec is None else prec
    B, N, D = dy2.shape
    Fc = pack1.cout
    dev = dy2.device
    sinks = sinks or {}
    i1, i2 = pack1.image(prec), pack2.image(prec)
    dz1 = torch.empty(B, N, 128, dtype=torch.float32, device=dev)
    dh = torch.empty(B, N, Fc, dtype=_H16[prec], device=dev)
    dg1 = torch.empty(B, N, 128, dtype=_H16[prec], device=dev)
    dg2 = torch.empty(B, N, 128, dtype=_H16[prec], device=dev)
    acc = {k: (sinks.get(k) if sinks.get(k) is not None else _zeros(arena, 128, device=dev)) for k in ('ln2_w', 'ln2_b', 'ln1_w', 'ln1_b')}
    dfilm = _zeros(arena, B, 256, device=dev) if film is not None else None
    datt = wt = None
    if out_pack is not None:
        if out_pack.cin != 128 or out_pack.cout != 128 or out_pack.taps != 1:
            raise ValueError('out_pack must be the (128, 128) out-projection')
        wt = out_pack.image(prec).bwd
        datt = torch.empty(B, N, 128, dtype=_H16[prec], device=dev)
    _log(pack1, ('ffpair', B * N, N, 128, Fc, 3))
    _fn('dx_ff_block_bwd', prec)(_p(dy2), _p(z2), _p(mean2), _p(rstd2), _p(ln2_w), _p(ln2_b), _p(film), 0 if film is None else film.stride(0),
                                 _p(dg2), _p(acc['ln2_w']), _p(acc['ln2_b']), _p(dfilm), 256, seed2, float(p2),
                                 _p(i2.bwd), _p(i1.bwd), _p(aux), _rows(aux), _p(dh), _rows(dh), _p(dz1), B, N, Fc, _p(lens), int(halo),
                                 _p(z1), _p(mean1), _p(rstd1), _p(ln1_w), _p(ln1_b), _p(dg1), _p(acc['ln1_w']), _p(acc['ln1_b']),
                                 seed1, float(p1), _p(wt), _p(datt), _p(seed_offset), _p(hmask), _stream())
    ret = lambda k: None if sinks.get(k) is not None else acc[k]
    return dz1, dh, dg1, dg2, dfilm, ret('ln2_w'), ret('ln2_b'), ret('ln1_w'), ret('ln1_b'), datt


class ZeroArena:
    """One zero-filled buffer handed out in 16-byte aligned slices: the many small accumulators (atomic targets) of one
    backward call cost a single memset launch instead of one each."""

    def __init__(self, device, n_floats):
        self.buf = torch.zeros(int(n_floats), dtype=torch.float32, device=device)
        self.off = 0
        self.requested = 0                        # floats asked for, including what did not fit

    @staticmethod
    def padded(n):
        return (int(n) + 3) // 4 * 4

    def take(self, *shape):
        n = 1
        for d in shape:
            n *= int(d)
        self.requested += self.padded(n)
        if self.off + n > self.buf.numel():
            return torch.zeros(*shape, dtype=torch.float32, device=self.buf.device)
        # an ALIAS of the buffer's storage, not a view of ``buf``: autograd's view / in-place bookkeeping (shared version counter,
        # "view created in no_grad mode ... base modified in place") must not couple the unrelated tensors carved out of one arena
        out = torch.empty(0, dtype=torch.float32, device=self.buf.device).set_(self.buf.untyped_storage(), self.off, tuple(int(d) for d in shape))
        self.off += self.padded(n)
        return out


def begin_step_arena(rt: Runtime, device) -> None:
    """One zero-filled buffer for every small accumulator / zero-initialised scratch tensor of the coming step (a training step had
    ~45 separate 4 us fill launches).  The owner calls ``end_step_arena`` when the step's backward has been issued: code running
    outside a step never sees a used arena."""
    rt.arena = ZeroArena(device, rt.arena_hint)


def end_step_arena(rt: Runtime) -> None:
    if rt.arena is not None:
        rt.arena_hint = max(rt.arena_hint, int(1.25 * rt.arena.requested))
        rt.arena = None


def _zeros(arena, *shape, device=None):
    return arena.take(*shape) if arena is not None else torch.zeros(*shape, dtype=torch.float32, device=device)


def conv_wgrad(dy, x, pack: PackedWeight, lens=None, halo=-1, bias=True, arena=None, w_sink=None, b_sink=None, prec=None, defer=False):
    """(dW, db): gradient w.r.t. the (Cout, Cin[, taps]) parameter in its own layout, and the bias gradient (column sums of
    dY, accumulated by the same launch).  ``w_sink`` / ``b_sink``: pre-zeroed ``.grad`` tensors to accumulate into directly
    (the corresponding return value is then None)."""
    B_, N_ = (1, x.shape[0]) if x.dim() == 2 else (x.shape[0], x.shape[1])
    # the kernel accumulates in the parameter's own (Cout, Cin, taps) layout: straight into the sink, or into a fresh gradient
    g = w_sink if w_sink is not None else _zeros(arena, *pack.weight.shape, device=x.device)
    db = b_sink if b_sink is not None else (_zeros(arena, pack.cout, device=x.device) if bias else None)
    _log(pack, ('wgrad', B_ * N_, N_, pack.cin, pack.cout, pack.taps))
    prec = prec or pack.rt.precision
    _check_h16(prec, dy, x)
    rt = pack.rt
    if (defer and rt.defer_wgrad and w_sink is not None and (b_sink is not None or not bias) and _half(prec) and pack.cin % 128 == 0
            and pack.cout % 8 == 0 and _rows(dy) % 8 == 0 and _rows(x) % 8 == 0):
        # queued: launched with up to 7 other layers of the same kind by flush_wgrads (the trainer flushes at the end of every backward phase)
        key = (pack.taps, _is_bf16(dy), _is_bf16(x), prec)
        rt.wgrad_queue.setdefault(key, []).append(((_p(dy), _p(x), _p(g), _p(db), _p(lens), _rows(dy), _rows(x), B_, N_, pack.cin, pack.cout,
                                                    int(halo), 0), (dy, x, g, db, lens)))
        return None, None
    _fn('dx_conv_wgrad', prec)(_p(dy), _rows(dy), _p(x), _rows(x), _p(g), B_, N_, pack.cin, pack.cout, pack.taps, _p(lens), int(halo),
                               _half(prec), _is_bf16(dy), _is_bf16(x), _p(db), _stream())
    return (None if w_sink is not None else g), (None if b_sink is not None else db)


class _WgradJob(ctypes.Structure):          # DxWgradJob of include/daft_exprt_hip.h
    _fields_ = [(n, ctypes.c_void_p) for n in ('dY', 'X', 'G', 'dbias', 'lens')] + \
               [(n, ctypes.c_int) for n in ('ldy', 'ldx', 'B', 'N', 'Cin', 'Cout', 'skip_halo', 'reserved')]


# layers per dx_conv_wgrad_batched launch (the library takes up to 32), by kernel size.  Measured in the C2 step: the k = 1 layers gain from
# one launch for all of them (4 launches 243 us -> 2 launches 172 us), the k = 3 layers do not (8 per launch: 668 us, 24 per launch: 687 us)
WGRAD_BATCH = {1: int(os.environ.get('DX_WGRAD_BATCH_K1', '32')), 3: int(os.environ.get('DX_WGRAD_BATCH_K3', '8'))}
WGRAD_BATCH_LOG = {}


def flush_wgrads(rt) -> int:
    """Launches the weight gradients queued on ``rt`` (conv_wgrad(defer=True)): per kind, up to 8 layers per launch
    (dx_conv_wgrad_batched).  Returns the number of launches.  Must run on the stream the producers ran on, before anything reads
    the gradients (the gradient exchange, the optimiser)."""
    launches = 0
    for (taps, dyh, xh, prec), jobs in rt.wgrad_queue.items():
        jobs.sort(key=lambda j: -j[0][7] * j[0][8])          # largest B * N first: the short symbol-level layers fill the tail of the launch
        for i in range(0, len(jobs), WGRAD_BATCH[taps]):
            part = jobs[i:i + WGRAD_BATCH[taps]]
            arr = (_WgradJob * len(part))(*[_WgradJob(*fields) for fields, _ in part])
            if _lib.TIMER is not None:       # diagnostic pricing of the launch (profiling.price): descriptors by array address
                WGRAD_BATCH_LOG[ctypes.addressof(arr)] = (arr, [fields for fields, _ in part])
            _fn('dx_conv_wgrad_batched', prec)(ctypes.addressof(arr), len(part), taps, dyh, xh, _stream())
            launches += 1
    rt.wgrad_queue.clear()
    return launches


def film_affine_fwd(gammas, betas, pm, nb):
    """(B, nb*C) predictor outputs + (2, nb) post-multipliers (or None) -> (nb, B, 2C) block-major FiLM parameters."""
    B, total = gammas.shape
    C = total // nb
    film = torch.empty(nb, B, 2 * C, dtype=torch.float32, device=gammas.device)
    lib().dx_film_affine_fwd(_p(gammas), _p(betas), _p(pm), _p(film), B, nb, C, _stream())
    return film


def film_affine_bwd(dfilms, gammas, betas, pm, arena=None):
    """``dfilms``: list of nb (B, 2C) gradients or None.  Returns (dgammas, dbetas, dpm or None)."""
    B, total = gammas.shape
    nb = len(dfilms)
    C = total // nb
    dfilms = [None if g is None else g.contiguous() for g in dfilms]
    ptrs = (ctypes.c_void_p * nb)(*[None if g is None else g.data_ptr() for g in dfilms])
    dg, db = torch.empty_like(gammas), torch.empty_like(betas)
    dpm = _zeros(arena, 2, nb, device=gammas.device) if pm is not None else None
    lib().dx_film_affine_bwd(ctypes.addressof(ptrs), _p(gammas), _p(betas), _p(pm), _p(dg), _p(db), _p(dpm), B, nb, C, _stream())
    return dg, db, dpm


def colsum(x, C=None):
    C = x.shape[-1] if C is None else C
    rows = x.numel() // x.shape[-1]
    out = torch.zeros(C, dtype=torch.float32, device=x.device)
    lib().dx_colsum(_p(x), _rows(x), _p(out), rows, C, _is_bf16(x), _stream())
    return out


# longest-first dispatch of the attention workgroups: measured neutral at C2 (28.2 vs 29.0 us: one resident round, no tail), so it is OFF by
# default and its two length_order launches per step are not issued; DX_ATTN_ORDER=1 turns it on (long-form / very uneven batches)
_ATTN_ORDER = os.environ.get('DX_ATTN_ORDER', '0') != '0'


def length_order(lens_i32):
    """Utterance indices sorted by length, longest first (device int32 [B]): the ``order`` argument of the attention wrappers."""
    order = torch.empty_like(lens_i32)
    lib().dx_length_order(_p(lens_i32), _p(order), lens_i32.shape[0], _stream())
    return order


def attention_fwd(qkv, lens, heads, seed, p_drop, prec=None, seed_offset=None, ctx_dtype=torch.float32, order=None):
    """``ctx_dtype``: storage type of the context (the 16-bit type of the operand mode, or float32).
    ``order``: optional ``length_order(lens)``: the longest utterances' workgroups are dispatched first."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    ctx = torch.empty(B, N, D, dtype=ctx_dtype, device=qkv.device)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device)
    prec = prec or DEFAULT.precision
    _check_h16(prec, qkv, ctx)
    _fn('dx_attention_fwd', prec)(_p(qkv), _rows(qkv), _p(lens), _p(ctx), D, _p(lse), B, N, heads, D, seed, _p(seed_offset), float(p_drop),
                           _half(prec), _is_bf16(qkv), _is_bf16(ctx), _p(order), _stream())
    return ctx, lse


def attention_bwd(qkv, ctx, dctx, lse, lens, heads, seed, p_drop, out_dtype=torch.float32, prec=None, seed_offset=None, order=None):
    B, N, D3 = qkv.shape
    D = D3 // 3
    if dctx.dtype != ctx.dtype:
        raise TypeError(f'ctx ({ctx.dtype}) and dctx ({dctx.dtype}) must be stored alike')
    dqkv = torch.empty(B, N, D3, dtype=out_dtype, device=qkv.device)
    delta = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device)
    prec = prec or DEFAULT.precision
    _fn('dx_attention_bwd', prec)(_p(qkv), _rows(qkv), _p(ctx), _p(dctx), _rows(dctx), _p(lse), _p(delta), _p(lens), _p(dqkv), _rows(dqkv),
                           B, N, heads, D, seed, _p(seed_offset), float(p_drop), _half(prec), _is_bf16(qkv), _is_bf16(dqkv), _is_bf16(ctx),
                           _p(order), _stream())
    return dqkv


def gemm_shadow(prec=None):
    """True when producers should also emit a bf16 copy of a 128-wide fp32 tensor that the next GEMM consumes (bf16 operand mode)."""
    return bool(_half(prec or DEFAULT.precision))


def ln_fwd(a, res, w, b, film, lens, *, seed_pre=0, p_pre=0.0, seed_post=0, p_post=0.0, halo=0, shadow=False, seed_offset=None, prec=None):
    """In place on ``a`` (becomes z = drop(a) + res).  Returns (y, mean, rstd) or, with ``shadow``, (y, mean, rstd, y_bf16)."""
    B, N, C = a.shape
    y = torch.empty_like(a)
    prec = prec or DEFAULT.precision
    y_h = torch.empty(B, N, C, dtype=_H16[prec], device=a.device) if shadow else None
    mean = torch.empty(B, N, dtype=torch.float32, device=a.device)
    rstd = torch.empty(B, N, dtype=torch.float32, device=a.device)
    _fn('dx_ln_fwd', prec)(_p(a), _p(res), _p(w), _p(b), _p(film), 0 if film is None else film.stride(0), _p(lens), int(halo), _p(y), _p(mean), _p(rstd),
                    B, N, C, seed_pre, float(p_pre), seed_post, float(p_post), _p(seed_offset), _is_bf16(a), _p(y_h), _stream())
    return (y, mean, rstd, y_h) if shadow else (y, mean, rstd)


def proj_ln_applies(x, pack: PackedWeight, prec) -> bool:
    """The fused out-projection + LayerNorm kernel serves the 16-bit operand modes at the FFT block's shape (128 -> 128, k = 1)."""
    return (prec in _H16 and x.dtype == _H16[prec] and x.dim() == 3 and x.shape[2] == 128 and pack.taps == 1 and pack.cin == 128
            and pack.cout == 128 and _PROJ_LN)


_PROJ_LN = os.environ.get('DX_PROJ_LN', '1') != '0'


_ATTN_PROJ_LN = os.environ.get('DX_ATTN_PROJ_LN', '1') != '0'


def attn_proj_ln_applies(qkv, heads, pack: PackedWeight, prec, order=None) -> bool:
    """Attention forward + out-projection + LayerNorm in one launch: the 16-bit modes at the FFT block's shape (2 heads x 64, 16-bit q/k/v)."""
    return (_ATTN_PROJ_LN and _PROJ_LN and prec in _H16 and qkv.dtype == _H16[prec] and qkv.dim() == 3 and qkv.shape[2] == 384 and heads == 2
            and order is None and pack.taps == 1 and pack.cin == 128 and pack.cout == 128)


def attn_proj_ln_fwd(qkv, lens, heads, seed, p_drop, pack: PackedWeight, proj_bias, res, w, b, film, *, seed_pre=0, p_pre=0.0, shadow=False,
                     seed_offset=None, prec=None):
    """attention_fwd (16-bit context) + proj_ln_fwd on it, one launch, the same bits.  Returns (ctx, lse, z, y, mean, rstd[, y_16bit])."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    prec = prec or pack.rt.precision
    _check_h16(prec, qkv)
    h16 = _H16[prec]
    ctx = torch.empty(B, N, D, dtype=h16, device=qkv.device)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device)
    z = torch.empty(B, N, D, dtype=torch.float32, device=qkv.device)
    y = torch.empty_like(z)
    y_h = torch.empty(B, N, D, dtype=h16, device=qkv.device) if shadow else None
    mean = torch.empty(B, N, dtype=torch.float32, device=qkv.device)
    rstd = torch.empty(B, N, dtype=torch.float32, device=qkv.device)
    img = pack.image(prec)
    _log(pack, ('gemm', B * N, N, pack.cin, pack.cout, 1))
    _fn('dx_attention_proj_ln_fwd', prec)(_p(qkv), _rows(qkv), _p(lens), _p(ctx), D, _p(lse), B, N, heads, D, seed, _p(seed_offset), float(p_drop),
                                          _p(img.fwd), _p(proj_bias), _p(z), _p(res), _p(w), _p(b), _p(film), 0 if film is None else film.stride(0),
                                          _p(y), _p(mean), _p(rstd), seed_pre, float(p_pre), _p(y_h), _stream())
    return (ctx, lse, z, y, mean, rstd, y_h) if shadow else (ctx, lse, z, y, mean, rstd)


def proj_ln_fwd(x, pack: PackedWeight, proj_bias, res, w, b, film, lens, *, seed_pre=0, p_pre=0.0, halo=0, shadow=False, seed_offset=None, prec=None):
    """z = dropout(x W^T + proj_bias) + res; y = mask(FiLM(LN(z))) in one launch.  Returns (z, y, mean, rstd[, y_16bit])."""
    B, N, C = x.shape
    prec = prec or pack.rt.precision
    _check_h16(prec, x)
    img = pack.image(prec)
    z = torch.empty(B, N, C, dtype=torch.float32, device=x.device)
    y = torch.empty_like(z)
    y_h = torch.empty(B, N, C, dtype=_H16[prec], device=x.device) if shadow else None
    mean = torch.empty(B, N, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, N, dtype=torch.float32, device=x.device)
    _log(pack, ('gemm', B * N, N, pack.cin, pack.cout, 1))
    _fn('dx_proj_ln_fwd', prec)(_p(x), _rows(x), _p(img.fwd), _p(proj_bias), _p(z), _p(res), _p(w), _p(b), _p(film), 0 if film is None else film.stride(0),
                                _p(lens), int(halo), _p(y), _p(mean), _p(rstd), B, N, seed_pre, float(p_pre), _p(seed_offset), _p(y_h), _stream())
    return (z, y, mean, rstd, y_h) if shadow else (z, y, mean, rstd)


def ln_bwd(dy, z, mean, rstd, w, b, film, lens, *, relu_mask=False, want_da=False, seed_pre=0, p_pre=0.0, seed_post=0, p_post=0.0,
           arena=None, w_sink=None, b_sink=None, halo=0, shadow=False, seed_offset=None, prec=None):
    """Returns (dz, da or None, dw, db, dfilm or None[, dg_bf16]); dw/db are None when accumulated straight into the given sinks."""
    B, N, C = z.shape
    dz = torch.empty_like(z)
    prec = prec or DEFAULT.precision
    dg_h = torch.empty(B, N, C, dtype=_H16[prec], device=z.device) if shadow else None
    da = torch.empty_like(z) if (want_da and not shadow) else None      # with a bf16 shadow the GEMMs read that copy instead
    dw = w_sink if w_sink is not None else _zeros(arena, C, device=z.device)
    db = b_sink if b_sink is not None else _zeros(arena, C, device=z.device)
    dfilm = _zeros(arena, B, 2 * C, device=z.device) if film is not None else None
    _fn('dx_ln_bwd', prec)(_p(dy), _p(z), _p(mean), _p(rstd), _p(w), _p(b), _p(film), 0 if film is None else film.stride(0), _p(lens), int(halo),
                    _p(dz), _p(da), _p(dw), _p(db), _p(dfilm), 2 * C, B, N, C, int(relu_mask),
                    seed_pre, float(p_pre), seed_post, float(p_post), _p(seed_offset), _is_bf16(z), _p(dg_h), _stream())
    out = (dz, da, (None if w_sink is not None else dw), (None if b_sink is not None else db), dfilm)
    return out + (dg_h,) if shadow else out


def add_pos(x, sym, emb, pe, lens):
    if emb is not None:
        B, N = sym.shape
        D = emb.shape[1]
    else:
        B, N, D = x.shape
    out = torch.empty(B, N, D, dtype=torch.float32, device=pe.device)
    lib().dx_add_pos(_p(x), _p(sym), _p(emb), _p(pe), _p(lens), _p(out), B, N, D, pe.shape[0], _stream())
    return out


def mask_rows(x, lens):
    B, N, C = x.shape
    out = torch.empty_like(x)
    lib().dx_mask_rows(_p(x), _p(lens), _p(out), B, N, C, _stream())
    return out


def embedding_bwd(dout, sym, lens, n_rows, arena=None, sink=None):
    """``sink``: the pre-zeroed ``.grad`` of the table to add into (the return value is then None)."""
    B, N, D = dout.shape
    demb = sink if sink is not None else _zeros(arena, n_rows, D, device=dout.device)
    lib().dx_embedding_bwd(_p(dout), _p(sym), _p(lens), _p(demb), B, N, D, _stream())
    return None if sink is not None else demb


def accent_sum(prenet, energy, pitch, we, be, wp, bp, pe, lens):
    B, N, D = prenet.shape
    out = torch.empty_like(prenet)
    lib().dx_accent_sum(_p(prenet), _p(energy), _p(pitch), _p(we), _p(be), _p(wp), _p(bp), _p(pe), _p(lens), _p(out), B, N, D, pe.shape[0], _stream())
    return out


def scalar_conv_wgrad(dout, s0, s1, lens, rowscale=None, sinks=None):
    """Gradients of Conv1d(1->128, k=3) weights/biases fed by the scalar streams s0 (and s1) given dout (B, N, 128).
    ``sinks``: (dw0, db0, dw1, db1) pre-zeroed ``.grad`` tensors (or None each) that the kernel accumulates into directly;
    the corresponding return value is then None."""
    B, N = s0.shape
    D = 128
    dev = s0.device
    sinks = sinks or (None, None, None, None)
    dw0 = sinks[0] if sinks[0] is not None else torch.zeros(D, 1, 3, dtype=torch.float32, device=dev)
    db0 = sinks[1] if sinks[1] is not None else torch.zeros(D, dtype=torch.float32, device=dev)
    dw1 = (sinks[2] if sinks[2] is not None else torch.zeros(D, 1, 3, dtype=torch.float32, device=dev)) if s1 is not None else None
    db1 = (sinks[3] if sinks[3] is not None else torch.zeros(D, dtype=torch.float32, device=dev)) if s1 is not None else None
    ldd = dout.stride(-2) if dout.dim() == 3 else 0  # a 1-D dout (D,) is broadcast over every row
    lib().dx_scalar_conv_wgrad(_p(dout), ldd, _p(rowscale), _p(s0), _p(s1), _p(lens), _p(dw0), _p(db0), _p(dw1), _p(db1), B, N, D, _stream())
    return tuple(None if sk is not None else t for sk, t in zip(sinks, (dw0, db0, dw1, db1)))


def mean_pool(x, lens, arena=None):
    B, N, C = x.shape
    out = _zeros(arena, B, C, device=x.device)
    lib().dx_mean_pool(_p(x), _p(lens), _p(out), B, N, C, _stream())
    return out


def mean_pool_bwd(dout, lens, N):
    B, C = dout.shape
    dx = torch.empty(B, N, C, dtype=torch.float32, device=dout.device)
    lib().dx_mean_pool_bwd(_p(dout), _p(lens), _p(dx), B, N, C, _stream())
    return dx


def transpose(x, add_to=None):
    """(B, R, C) -> (B, C, R), contiguous.  ``add_to``: a contiguous (B, C, R) tensor that the transpose is ADDED to in place (returned)."""
    B, R, C = x.shape
    out = add_to if add_to is not None else torch.empty(B, C, R, dtype=torch.float32, device=x.device)
    lib().dx_transpose(_p(x), _p(out), B, R, C, int(add_to is not None), _stream())
    return out


def l2_normalize(x):
    y = torch.empty_like(x)
    lib().dx_l2_normalize(_p(x), _p(y), x.shape[0], x.shape[1], _stream())
    return y


def cross_entropy(logits, target):
    B, S = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    lib().dx_cross_entropy(_p(logits), _p(target), _p(loss), _p(dlogits), B, S, _stream())
    return loss, dlogits


def duration_scan(dur_int):
    B, L = dur_int.shape
    mu = torch.empty(B, L, dtype=torch.float32, device=dur_int.device)
    totals = torch.empty(B, dtype=torch.long, device=dur_int.device)
    lib().dx_duration_scan(_p(dur_int), _p(mu), _p(totals), B, L, _stream())
    return mu, totals


def upsample_prep(enc, dur, energy, pitch, wd, bd, we, be, wp, bp, wr, br, lens):
    B, L, D = enc.shape
    xs = torch.empty_like(enc)
    z = torch.empty(B, L, dtype=torch.float32, device=enc.device)
    sigma = torch.empty(B, L, dtype=torch.float32, device=enc.device)
    lib().dx_upsample_prep(_p(enc), _p(dur), _p(energy), _p(pitch), _p(wd), _p(bd), _p(we), _p(be), _p(wp), _p(bp), _p(wr), _p(br),
                           _p(lens), _p(xs), _p(z), _p(sigma), B, L, D, _stream())
    return xs, z, sigma


def upsample_fwd(xs, mu, sigma, lens, T):
    B, L, D = xs.shape
    weights = torch.empty(B, L, T, dtype=torch.float32, device=xs.device)
    xup = torch.empty(B, T, D, dtype=torch.float32, device=xs.device)
    lib().dx_upsample_fwd(_p(xs), _p(mu), _p(sigma), _p(lens), _p(weights), _p(xup), B, L, T, D, _stream())
    return xup, weights


def upsample_bwd(dxup, xs, mu, sigma, weights, lens, arena=None):
    B, L, D = xs.shape
    T = weights.shape[2]
    dxs = _zeros(arena, B, L, D, device=xs.device)
    dsigma = _zeros(arena, B, L, device=xs.device)
    lib().dx_upsample_bwd(_p(dxup), _p(xs), _p(mu), _p(sigma), _p(weights), _p(lens), _p(dxs), _p(dsigma), B, L, T, D, _stream())
    return dxs, dsigma


def upsample_sym_bwd(dxs, dsigma, xs, z, dur, lens, wd, bd, wr, dwr_sink=None, dbr_sink=None):
    """``dwr_sink`` / ``dbr_sink``: pre-zeroed ``.grad`` tensors to accumulate into (the returned dwr / dbr is then None)."""
    B, L, D = xs.shape
    dxs_out = torch.empty_like(xs)
    dz = torch.empty(B, L, dtype=torch.float32, device=xs.device)
    dwr = dwr_sink if dwr_sink is not None else torch.zeros(D, dtype=torch.float32, device=xs.device)
    dbr = dbr_sink if dbr_sink is not None else torch.zeros(1, dtype=torch.float32, device=xs.device)
    lib().dx_upsample_sym_bwd(_p(dxs), _p(dsigma), _p(xs), _p(z), _p(dur), _p(lens), _p(wd), _p(bd), _p(wr), _p(dxs_out), _p(dz), _p(dwr), _p(dbr),
                              B, L, D, _stream())
    return dxs_out, dz, (None if dwr_sink is not None else dwr), (None if dbr_sink is not None else dbr)


def mel_stats(mel_pred, mel_target, arena=None):
    B, M, T = mel_pred.shape
    dev = mel_pred.device
    ep = torch.empty(B, T, dtype=torch.float32, device=dev)
    et = torch.empty(B, T, dtype=torch.float32, device=dev)
    sums = _zeros(arena, 2, B, device=dev)
    lib().dx_mel_stats(_p(mel_pred), _p(mel_target), _p(ep), _p(et), _p(sums[0]), _p(sums[1]), B, M, T, _stream())
    return ep, et, sums


def energy_diff(ep, et, lens, arena=None):
    B, T = ep.shape
    des = torch.empty_like(ep)
    esum = _zeros(arena, 1, device=ep.device)
    lib().dx_energy_diff(_p(ep), _p(et), _p(lens), _p(des), _p(esum), B, T, _stream())
    return des, esum


def mel_grad(mel_pred, mel_target, ep, des, lens, c_l1, c_l2, c_e, e_per_total=False):
    B, M, T = mel_pred.shape
    dmel = torch.empty_like(mel_pred)
    lib().dx_mel_grad(_p(mel_pred), _p(mel_target), _p(ep), _p(des), _p(lens), float(c_l1), float(c_l2), float(c_e), int(e_per_total), _p(dmel),
                      B, M, T, _stream())
    return dmel


def loss_finalize(ce, dlogits, spk_w, pm, pmw, sums, lens, M, msw, esum, ecw, psum, pcw, grad_scale=1.0):
    """-> (terms[7], total[1], d_spk or None, d_pm or None); ``spk_w``: python float or device scalar tensor"""
    dev = sums.device
    B = sums.shape[1]
    out = torch.empty(8, dtype=torch.float32, device=dev)
    d_spk = torch.empty_like(dlogits) if dlogits is not None else None
    d_pm = torch.empty_like(pm) if pm is not None else None
    w_dev = spk_w if torch.is_tensor(spk_w) else None
    lib().dx_loss_finalize(_p(ce), _p(w_dev), 0.0 if w_dev is not None else float(spk_w), _p(dlogits), _p(d_spk), 0 if dlogits is None else dlogits.numel(),
                           _p(pm), _p(d_pm), 0 if pm is None else pm.numel(), float(pmw), _p(sums[0]), _p(sums[1]), _p(lens), B, M, float(msw),
                           _p(esum), float(ecw), _p(psum), float(pcw), _p(out), _p(out[7:]), float(grad_scale), _stream())
    return out[:7], out[7], d_spk, d_pm


def _frame_stride(t):
    """(B, T) contiguous -> 1; (B, T, C) contiguous -> C (channel 0 of every row is meant)."""
    return 1 if t.dim() == 2 else t.shape[2]


def pitch_mse(pp, gt, lens, arena=None):
    """``pp``: (B, T), or the (B, T, C) output of the predictor's last convolution (channel 0 is read in place)."""
    B, T = gt.shape
    sums = _zeros(arena, 2, device=gt.device)
    lib().dx_pitch_mse(_p(pp), _frame_stride(pp), _p(gt), _p(lens), _p(sums), B, T, _stream())
    return sums


def pitch_grad(pp, gt, lens, sums, scale, out=None):
    """``out``: optional (B, T, C) tensor whose channel 0 receives the gradient (the other channels are left as they are)."""
    B, T = gt.shape
    dpp = out if out is not None else torch.empty(B, T, dtype=torch.float32, device=gt.device)
    lib().dx_pitch_grad(_p(pp), _frame_stride(pp), _p(gt), _p(lens), _p(sums), float(scale), _p(dpp), _frame_stride(dpp), B, T, _stream())
    return dpp


def pitch_chain_applies(layers, mel, prec) -> bool:
    """The fused frozen-predictor launches (csrc/dx_pitch.hip) cover the reference architecture in the 16-bit modes: three 256-wide k = 3
    convolutions with folded BatchNorm and a k = 3 convolution to one channel, 64 < n_mel <= 96 (the reference: 80)."""
    if not _half(prec) or len(layers) != 4 or mel.dim() != 3 or not (64 < mel.shape[1] <= 96) or mel.shape[0] > 1000:
        return False
    shapes = [tuple(l['pack'].weight.shape) for l in layers]
    M = mel.shape[1]
    return (shapes[0] == (256, M, 3) and shapes[1] == (256, 256, 3) and shapes[2] == (256, 256, 3) and shapes[3][1:] == (256, 3)
            and all(l['scale'] is not None for l in layers[:3]))


def pitch_chain_fwd(mel, layers, lens, prec, arena=None):
    """(pp (B, T), masks) = the frozen pitch predictor on mel (B, n_mel, T) fp32 in one launch; ``masks`` feeds pitch_chain_bwd."""
    B, M, T = mel.shape
    pp = _zeros(arena, B, T, device=mel.device)          # tokens >= len are never written (and never read unmasked)
    masks = torch.empty(B, T, 3, 8, dtype=torch.int32, device=mel.device)
    img = [l['pack'].image(prec) for l in layers[:3]]
    _fn('dx_pitch_chain_fwd', prec)(_p(mel), B, M, T, _p(lens), _p(img[0].fwd), _p(img[1].fwd), _p(img[2].fwd),
                                    _p(layers[0]['b']), _p(layers[1]['b']), _p(layers[2]['b']),
                                    _p(layers[0]['scale']), _p(layers[1]['scale']), _p(layers[2]['scale']),
                                    _p(layers[0]['shift']), _p(layers[1]['shift']), _p(layers[2]['shift']),
                                    _p(layers[3]['w']), float(layers[3]['b3']), _p(pp), _p(masks), _stream())
    return pp, masks


def pitch_chain_bwd(dpp, masks, layers, lens, prec, dmel):
    """dmel (B, n_mel, T) += the gradient of the frozen predictor's output with respect to its input mel, from dpp (B, T)."""
    B, M, T = dmel.shape
    img = [l['pack'].image(prec) for l in layers[:3]]
    _fn('dx_pitch_chain_bwd', prec)(_p(dpp), B, M, T, _p(lens), _p(img[0].bwd), _p(img[1].bwd), _p(img[2].bwd),
                                    _p(layers[0]['scale']), _p(layers[1]['scale']), _p(layers[2]['scale']),
                                    _p(layers[3]['w']), _p(masks), _p(dmel), _stream())
    return dmel


def relu_bwd(dy, y):
    out = torch.empty_like(dy)
    lib().dx_relu_bwd(_p(dy), _p(y), _p(out), dy.numel(), _stream())
    return out


def channel_affine(x, scale, shift, prec=None):
    out = torch.empty_like(x)
    fn = _fn('dx_channel_affine', prec or DEFAULT.precision) if _is_bf16(x) else lib().dx_channel_affine
    fn(_p(x), _p(scale), _p(shift), _p(out), x.numel() // x.shape[-1], x.shape[-1], _is_bf16(x), _stream())
    return out
