"""Deterministic synthetic weights and batches.

There are no checkpoints or datasets in the reference tree,
so benchmarks and parity fixtures run on seeded synthetic data.  Everything
here is a pure function of (name, shape, seed) on the torch CPU generator, so
the build container (which writes the golden fixtures with the reference
model) and the GPU box (which only has this repository) regenerate the same
tensors without shipping 57 MB of weights.

Batch layout follows the reference collate function
(src/daft_exprt/data_loader.py:207-287) and the distributions fixed in
SURVEY.md §8(d).
"""
from __future__ import annotations

import math
import zlib

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synthetic_tensor(name: str, shape, seed: int, dtype=torch.float32) -> torch.Tensor:
    """One parameter / buffer, scaled so that activations stay O(1) through 12 blocks."""
    shape = tuple(shape)
    g = _gen(name, seed)
    if name.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith('running_var'):
        return 1.0 + 0.2 * torch.rand(shape, generator=g, dtype=dtype)
    if name.endswith('running_mean'):
        return 0.1 * torch.randn(shape, generator=g, dtype=dtype)
    if name.endswith('post_multipliers'):
        return 0.5 * torch.randn(shape, generator=g, dtype=dtype)
    if name.endswith('weight_g'):
        return 0.75 + 0.5 * torch.rand(shape, generator=g, dtype=dtype)
    if len(shape) <= 1:
        if name.endswith('weight'):  # LayerNorm / BatchNorm scale
            return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=dtype)
        return 0.05 * torch.randn(shape, generator=g, dtype=dtype)
    receptive = 1
    for s in shape[2:]:
        receptive *= s
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    std = math.sqrt(2.0 / (fan_in + fan_out))
    return std * torch.randn(shape, generator=g, dtype=dtype)


def synthetic_state_dict(shapes: dict, seed: int = 1234) -> dict:
    """``shapes``: name -> shape (e.g. ``{k: v.shape for k, v in model.state_dict().items()}``)."""
    return {name: synthetic_tensor(name, shape, seed) for name, shape in shapes.items()}


def synthetic_batch(batch_size, sym_len_range, dur_range=(2, 12), seed=1234, n_speakers=2, n_symbols=76,
                    n_mels=80, spk_dim=192, zero_dur_frac=0.0, sym_lengths=None, durations_int=None):
    """Returns the reference 14-tuple (data_loader.py:276-287), CPU tensors.

    ``sym_lengths`` / ``durations_int`` override the random draw (used by the
    padding-halo fixtures that need exact length patterns).
    """
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    lo, hi = sym_len_range
    if sym_lengths is None:
        lens = torch.randint(lo, hi + 1, (batch_size,), generator=g)
        lens[0] = hi
        lens, _ = torch.sort(lens, descending=True)
    else:
        lens = torch.as_tensor(sym_lengths, dtype=torch.long)
        batch_size = lens.numel()
    L = int(lens.max())
    sym_mask = torch.arange(L)[None, :] < lens[:, None]
    symbols = torch.randint(1, n_symbols, (batch_size, L), generator=g) * sym_mask
    if durations_int is None:
        dur_int = torch.randint(dur_range[0], dur_range[1] + 1, (batch_size, L), generator=g)
        if zero_dur_frac > 0:
            drop = torch.rand((batch_size, L), generator=g) < zero_dur_frac
            drop[:, 0] = False
            dur_int = dur_int.masked_fill(drop, 0)
        dur_int = dur_int * sym_mask
    else:
        dur_int = torch.as_tensor(durations_int, dtype=torch.long) * sym_mask
    out_lens = dur_int.sum(dim=1)
    T = int(out_lens.max())
    frm_mask = torch.arange(T)[None, :] < out_lens[:, None]
    dur_float = dur_int.float() * (256.0 / 22050.0)

    def _prosody(shape, zero_frac, mask, absolute=False):
        v = torch.randn(shape, generator=g)
        if absolute:
            v = v.abs()
        z = torch.rand(shape, generator=g) < zero_frac
        return v.masked_fill(z, 0.0) * mask

    sym_energy = _prosody((batch_size, L), 0.25, sym_mask)
    sym_pitch = _prosody((batch_size, L), 0.25, sym_mask)
    frm_energy = _prosody((batch_size, T), 0.0, frm_mask, absolute=True)
    frm_pitch = _prosody((batch_size, T), 0.30, frm_mask)
    mel = (-5.0 + 2.0 * torch.randn((batch_size, n_mels, T), generator=g)).clamp_(math.log(1e-5), 2.0)
    mel = mel * frm_mask[:, None, :]
    speaker_ids = torch.randint(0, max(n_speakers - 1, 1), (batch_size,), generator=g)
    spk_embs = torch.randn((batch_size, spk_dim), generator=g)
    feature_dirs = ['synthetic'] * batch_size
    feature_files = [f'utt_{seed}_{i:04d}' for i in range(batch_size)]
    return (symbols.long(), dur_float, dur_int.long(), sym_energy, sym_pitch, lens.long(),
            frm_energy, frm_pitch, mel, out_lens.long(), speaker_ids.long(),
            feature_dirs, feature_files, spk_embs)


# BASELINE.json configs -> synthetic batch arguments (C1..C5)
CONFIGS = {
    'C1': dict(batch_size=4, sym_len_range=(50, 100), seed=1235),
    'C2': dict(batch_size=48, sym_len_range=(50, 120), seed=1236),
    'C3': dict(batch_size=48, sym_len_range=(50, 120), seed=1237, n_speakers=12),
    'C4': dict(batch_size=256, sym_len_range=(60, 100), seed=1238),
    'C5': dict(batch_size=8, sym_len_range=(400, 500), dur_range=(4, 12), seed=1239),
}


def synthetic_inference_batch(batch_size=256, sym_len_range=(60, 100), seed=1238, n_symbols=76, spk_dim=192, accent_dim=128,
                              n_speakers=2):
    """BASELINE.json config 4 (C4) inputs of ``DaftExprt.inference`` (reference model.py:1026-1114), CPU tensors:
    (inputs 6-tuple, external_prosody dict, external_embeddings (B, 192), external_accent_emb (B, 128)).
    Durations are seconds in [0.05, 0.14) per valid symbol (4-12 frames at hop 256 / 22050 Hz): T_max ~ 800 at L = 100."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    lo, hi = sym_len_range
    lens = torch.randint(lo, hi + 1, (batch_size,), generator=g)
    lens[0] = hi
    lens, _ = torch.sort(lens, descending=True)
    L = int(lens.max())
    valid = torch.arange(L)[None, :] < lens[:, None]
    symbols = torch.randint(1, n_symbols, (batch_size, L), generator=g) * valid
    dur = (0.05 + 0.09 * torch.rand(batch_size, L, generator=g)) * valid
    energy = torch.randn(batch_size, L, generator=g) * valid
    pitch = torch.randn(batch_size, L, generator=g).masked_fill(torch.rand(batch_size, L, generator=g) < 0.25, 0.0) * valid
    inputs = (symbols.long(), torch.ones(batch_size, L), torch.ones(batch_size, L), 0.5 * torch.ones(batch_size, L), lens.long(),
              torch.randint(0, max(n_speakers - 1, 1), (batch_size,), generator=g).long())
    prosody = {'duration_preds': dur, 'durations_int': torch.zeros(batch_size, L, dtype=torch.long), 'energy_preds': energy, 'pitch_preds': pitch}
    return inputs, prosody, torch.randn(batch_size, spk_dim, generator=g), 0.3 * torch.randn(batch_size, accent_dim, generator=g)
