"""Per-kernel roofline accounting for bench.py and tools/ (measurement support: nothing here computes anything on the path).

``_lib.set_timer(list)`` makes every C-ABI launch record (entry point, its C arguments by name, start event, stop event) on the
stream the kernels run on.  ``price()`` turns one record into ALGORITHMIC work -- FLOPs for the MFMA-bound kernels, HBM bytes for
the memory-bound ones -- following SURVEY.md §8(d): valid tokens only (padding the kernels compute or skip is not credited),
multiply-add = 2, every tensor the operation must read or write counted once at its storage width.  ``summarize()`` groups the
records of a step by kernel and divides by the measured durations.

Peaks: /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters (HBM3E 8.0 TB/s spec; dense MFMA 2.5 PFLOP/s bf16, 157.3
TFLOP/s f32-input).
"""
from __future__ import annotations

from collections import OrderedDict

PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}
PEAK_HBM_GBS = 8000.0


class Geometry:
    """valid rows and sum of squared lengths per padded axis length N of the batch (frames and symbols)."""

    def __init__(self, axes):
        """``axes``: iterable of 1-D integer sequences of per-utterance lengths (one per axis)."""
        self.by_n = {}
        for lens in axes:
            lens = [int(v) for v in lens]
            self.by_n[(len(lens), max(lens))] = (sum(lens), sum(v * v for v in lens))

    def rows(self, B, N, lens_ptr=True):
        if not lens_ptr:
            return B * N
        return self.by_n.get((B, N), (B * N, B * N * N))[0]

    def pairs(self, B, N):
        return self.by_n.get((B, N), (B * N, B * N * N))[1]


def _conv_route(a):
    """Mirror of the kernel choice in csrc/dx_gemm.hip dx_conv_gemm (for labelling only)."""
    if not a['bf16']:
        return 'conv_gemm_kernel<f32>'
    cinp = (a['Cin'] + 63) // 64 * 64
    if cinp >= 256 and a['Cin'] % 64 == 0:
        return f"conv_dk_kernel<{a['taps']}>"
    if cinp == 128 and a['B'] * ((a['N'] + 127) // 128) >= 64:
        return f"conv_ws_kernel<{a['taps']}>"
    return 'conv_gemm_kernel<bf16>'


def price(name, a, geom: Geometry):
    """-> (kernel label, bound, algorithmic FLOPs, algorithmic HBM bytes); bytes / FLOPs may be None when not modelled."""
    g = a.get
    has = lambda k: bool(g(k))
    if name.endswith('_f16'):                            # the fp16 twins (include/daft_exprt_hip.h): same kernels, same work
        name = name[:-4]
    if name == 'dx_conv_gemm':
        rows = geom.rows(a['B'], a['N'], has('lens'))
        xb, yb, ab = (2 if a['x_bf16'] else 4), (2 if a['y_bf16'] else 4), (2 if a['aux_bf16'] else 4)
        wb = 2 if a['bf16'] else 4
        byt = rows * (a['Cin'] * xb + a['Cout'] * yb * (2 if a['accumulate'] else 1) + (a['Cout'] * ab if has('relu_aux') else 0)) \
            + a['taps'] * a['Cin'] * a['Cout'] * wb
        return _conv_route(a), 'mfma', 2.0 * a['taps'] * a['Cin'] * a['Cout'] * rows, byt
    if name == 'dx_ff_pair':
        rows = geom.rows(a['B'], a['N'], has('lens'))
        F = a['F']
        byt = rows * (128 * 2 + F * 2 + 128 * 4 * (2 if a['accumulate'] else 1) + (F * 2 if has('aux') else 0)) + 2 * 3 * 128 * F * 2
        return 'ff_pair_kernel<bwd>' if has('aux') else 'ff_pair_kernel<fwd>', 'mfma', 2.0 * 2 * 3 * 128 * F * rows, byt
    if name == 'dx_ff_pair_ln':                          # the forward pair + the LayerNorm epilogue: residual read, z and y written
        rows = geom.rows(a['B'], a['N'], has('lens'))
        F = a['F']
        byt = rows * (128 * 2 + F * 2 + 128 * 4 * 3 + 8) + 2 * 3 * 128 * F * 2
        return 'ff_pair_kernel<fwd>', 'mfma', 2.0 * 2 * 3 * 128 * F * rows, byt
    if name == 'dx_ff_pair_ln_qkv':                      # ... + the next block's in-projection (128 -> 384) on the normalised tile: 16-bit qkv written
        rows = geom.rows(a['B'], a['N'], has('lens'))
        F = a['F']
        byt = rows * (128 * 2 + F * 2 + 128 * 4 * 3 + 8 + 384 * 2) + 2 * 3 * 128 * F * 2 + 384 * 128 * 2
        return 'ff_pair_kernel<fwd>', 'mfma', (2.0 * 2 * 3 * 128 * F + 2.0 * 128 * 384) * rows, byt
    if name == 'dx_ff_pair_lnbwd':                       # the input-gradient pair + the LayerNorm-backward epilogue: z read, dz and the 16-bit copy written
        rows = geom.rows(a['B'], a['N'], has('lens'))
        F = a['F']
        byt = rows * (128 * 2 + 2 * F * 2 + 128 * 4 * 3 + 128 * 2 + 8) + 2 * 3 * 128 * F * 2
        return 'ff_pair_kernel<bwd>', 'mfma', 2.0 * 2 * 3 * 128 * F * rows, byt
    if name == 'dx_ff_block_bwd':                        # LN2-backward prologue + input-gradient pair + LN1-backward epilogue
        rows = geom.rows(a['B'], a['N'], has('lens'))
        F = a['F']
        byt = rows * (2 * 128 * 4 + 2 * 128 * 4 + 2 * 128 * 2 + 2 * F * 2 + 128 * 4 + 16) + 2 * 3 * 128 * F * 2   # dY2, z2, z1 read; dz2 + dz1 written; two 16-bit copies; aux + H
        return 'ff_pair_kernel<bwd>', 'mfma', 2.0 * 2 * 3 * 128 * F * rows, byt
    if name == 'dx_conv_wgrad':
        rows = geom.rows(a['B'], a['N'], has('lens'))
        byt = rows * (a['Cout'] * (2 if a['dy_bf16'] else 4) + a['Cin'] * (2 if a['x_bf16'] else 4)) + a['taps'] * a['Cin'] * a['Cout'] * 4
        wide = a['Cin'] % 128 == 0 and -(-a['Cout'] // 128) * -(-a['Cin'] // 64) >= 64      # the library's choice of the 128 x 128 tile (dx_gemm.hip)
        label = (f"wgrad_bf16_kernel<{a['taps']}{',wide' if wide else ''}>" if a['bf16'] and a['Cin'] % 8 == 0 and a['Cout'] % 8 == 0
                 else f"wgrad_kernel<{a['taps']}>")
        return label, 'mfma', 2.0 * a['taps'] * a['Cin'] * a['Cout'] * rows, byt
    if name == 'dx_conv_wgrad_batched':
        from . import ops
        jobs = ops.WGRAD_BATCH_LOG.get(a['jobs'], (None, []))[1]       # the descriptors of this launch (kept while a timer is installed)
        flops = byt = 0
        for (_dy, _x, _g, _db, lens, _ldy, _ldx, B, N, Cin, Cout, _halo, _r) in jobs:
            rows = geom.rows(B, N, lens is not None)
            flops += 2.0 * a['taps'] * Cin * Cout * rows
            byt += rows * (Cout * (2 if a['dy_bf16'] else 4) + Cin * (2 if a['x_bf16'] else 4)) + a['taps'] * Cin * Cout * 4
        return f"wgrad_bf16_kernel<{a['taps']},wide>", 'mfma', (flops or None), (byt or None)      # batched launches use the wide tile
    if name in ('dx_attention_fwd', 'dx_attention_bwd'):
        rows, pairs = geom.rows(a['B'], a['N']), geom.pairs(a['B'], a['N'])
        D, H = a['D'], a['H']
        qb = 2 if a['qkv_bf16'] else 4
        if name == 'dx_attention_fwd':                       # S = QK^T and O = PV: 2 products of 2 N^2 hd per head
            return 'attn_fwd', 'mfma', 4.0 * pairs * D, rows * (3 * D * qb + D * 4 + H * 4)
        gb = 2 if a['dqkv_bf16'] else 4                      # dQ kernel: S, dP, dQ; dK/dV kernel: S, dP, dV, dK: 7 products
        return 'attn_bwd (dq + dkv)', 'mfma', 14.0 * pairs * D, rows * (2 * 3 * D * qb + 2 * D * 4 + 3 * D * gb + 2 * H * 4)
    if name in ('dx_pitch_chain_fwd', 'dx_pitch_chain_bwd'):      # the frozen predictor: 3 taps x (M x 256 + 2 x 256 x 256) MACs per token + the one-channel layer
        rows = geom.rows(a['B'], a['T'], True)
        flops = 2.0 * 3 * (a['M'] * 256 + 2 * 256 * 256 + 256) * rows
        byt = rows * (a['M'] * 4 + 4 + 96 + (a['M'] * 8 if name.endswith('bwd') else 0))       # mel / pp (dpp) / sign bits; backward: dmel read + written
        return ('pitch_chain<fwd>' if name.endswith('fwd') else 'pitch_chain<bwd>'), 'mfma', flops, byt
    if name == 'dx_attention_proj_ln_fwd':               # attn_fwd + proj_ln_fwd in one launch: both products, the 128 x 128 projection, the row pass's bytes
        rows, pairs = geom.rows(a['B'], a['N']), geom.pairs(a['B'], a['N'])
        D, H = a['D'], a['H']
        byt = rows * (3 * D * 2 + D * 2 + H * 4) + rows * (128 * (4 + 4 + (4 if has('res') else 0) + (2 if has('y_bf16_copy') else 0)) + 8) + 128 * 128 * 2
        return 'attn_fwd+proj_ln', 'mfma', 4.0 * pairs * D + 2.0 * rows * 128 * 128, byt
    if name == 'dx_ln_fwd':
        rows = geom.rows(a['B'], a['N'], has('lens'))
        C, ab = a['C'], (2 if a['io_bf16'] else 4)
        byt = rows * (C * (2 * ab + 4 + (4 if has('res') else 0) + (2 if has('y_bf16_copy') else 0)) + 8)
        return f'ln_fwd_kernel<{C}>', 'hbm', None, byt
    if name == 'dx_proj_ln_fwd':                         # X (16-bit) read; z, y written, residual read (fp32); 16-bit copy of y; the 32 KB weight pack
        rows = geom.rows(a['B'], a['N'], has('lens'))
        byt = rows * (128 * (2 + 4 + 4 + (4 if has('res') else 0) + (2 if has('y_bf16_copy') else 0)) + 8) + 128 * 128 * 2
        return 'proj_ln_fwd', 'hbm', None, byt
    if name == 'dx_ln_bwd':
        rows = geom.rows(a['B'], a['N'], has('lens'))
        C, zb = a['C'], (2 if a['io_bf16'] else 4)
        byt = rows * (C * (3 * zb + (4 if has('da') else 0) + (2 if has('dg_bf16_copy') else 0)) + 8)   # dy, z read; dz (+ da, bf16 copy) written
        return f'ln_bwd_kernel<{C}>', 'hbm', None, byt
    if name == 'dx_upsample_fwd':
        B, L, T, D = a['B'], a['L'], a['T'], a['D']
        return 'upsample_fwd', 'hbm', None, B * L * D * 4 + 3 * B * L * 4 + B * T * D * 4 + B * L * T * 4
    if name == 'dx_upsample_bwd':
        B, L, T, D = a['B'], a['L'], a['T'], a['D']
        return 'upsample_bwd (dsigma + dxs)', 'hbm', None, B * T * D * 4 + B * L * T * 4 + 2 * B * L * D * 4 + 3 * B * L * 4
    if name in ('dx_upsample_prep', 'dx_upsample_sym_bwd'):
        return name[3:], 'hbm', None, 2 * a['B'] * a['L'] * a['D'] * 4 + 6 * a['B'] * a['L'] * 4
    if name == 'dx_adam_step':
        return 'adam_kernel', 'hbm', None, a['n'] * 28
    if name == 'dx_sumsq':
        return 'sumsq_kernel', 'hbm', None, a['n'] * 4
    if name == 'dx_mel_stats':
        return 'mel_stats', 'hbm', None, 2 * a['B'] * a['M'] * a['T'] * 4
    if name == 'dx_mel_grad':
        return 'mel_grad', 'hbm', None, 3 * a['B'] * a['M'] * a['T'] * 4
    if name == 'dx_add_pos':
        rows = geom.rows(a['B'], a['N'])
        return 'add_pos', 'hbm', None, rows * a['D'] * 4 * (2 if has('x') else 1)
    if name == 'dx_accent_sum':
        return 'accent_sum', 'hbm', None, geom.rows(a['B'], a['N']) * (2 * a['D'] * 4 + 8)
    if name == 'dx_scalar_conv_wgrad':
        return 'scalar_conv_wgrad', 'hbm', None, geom.rows(a['B'], a['N']) * (a['D'] * 4 if a['ldd'] else 0) + geom.rows(a['B'], a['N']) * 8
    if name == 'dx_transpose':
        return 'transpose', 'hbm', None, 2 * a['B'] * a['R'] * a['Cc'] * 4
    if name == 'dx_mask_rows':
        return 'mask_rows', 'hbm', None, 2 * geom.rows(a['B'], a['N']) * a['C'] * 4
    if name == 'dx_mean_pool':
        return 'mean_pool', 'hbm', None, geom.rows(a['B'], a['N']) * a['C'] * 4
    if name == 'dx_mean_pool_bwd':
        return 'mean_pool_bwd', 'hbm', None, geom.rows(a['B'], a['N']) * a['C'] * 4
    if name == 'dx_channel_affine':
        return 'channel_affine', 'hbm', None, 2 * a['rows'] * a['C'] * (2 if a.get('io_bf16') else 4)
    if name == 'dx_relu_bwd':
        return 'relu_bwd', 'hbm', None, 3 * a['n'] * 4
    if name == 'dx_colsum':
        return 'colsum', 'hbm', None, a['rows'] * a['C'] * (2 if a['x_bf16'] else 4)
    return name[3:], 'hbm', None, None


def summarize(records, geom: Geometry, precision: str):
    """records of ONE step -> OrderedDict kernel label -> {launches, total_us, avg_us, bound, achieved, unit, peak, frac, ...},
    sorted by total time.  Events must have completed (synchronise first)."""
    rows = {}
    for name, args, ev0, ev1 in records:
        label, bound, flops, byt = price(name, args, geom)
        e = rows.setdefault(label, {'launches': 0, 'total_us': 0.0, 'bound': bound, 'flops': 0.0, 'bytes': 0.0, 'unpriced': 0})
        e['launches'] += 1
        e['total_us'] += ev0.elapsed_time(ev1) * 1e3
        if bound == 'mfma' and flops is not None:
            e['flops'] += flops
        if byt is not None:
            e['bytes'] += byt
        else:
            e['unpriced'] += 1
    out = OrderedDict()
    for label, e in sorted(rows.items(), key=lambda kv: -kv[1]['total_us']):
        t = e['total_us'] * 1e-6
        r = {'launches': e['launches'], 'total_us': round(e['total_us'], 1), 'avg_us': round(e['total_us'] / e['launches'], 2), 'bound': e['bound']}
        if e['bound'] == 'mfma':
            peak = PEAK_TFLOPS['f32' if '<f32>' in label or (precision == 'f32') else 'bf16']
            r.update(achieved=round(e['flops'] / t / 1e12, 2), unit='TFLOP/s', peak=peak, frac=round(e['flops'] / t / 1e12 / peak, 4),
                     algorithmic_bytes_per_launch=int(e['bytes'] / e['launches']))
        elif e['unpriced'] == 0:
            r.update(achieved=round(e['bytes'] / t / 1e9, 1), unit='GB/s', peak=PEAK_HBM_GBS, frac=round(e['bytes'] / t / 1e9 / PEAK_HBM_GBS, 4),
                     algorithmic_bytes_per_launch=int(e['bytes'] / e['launches']))
        out[label] = r
    return out
