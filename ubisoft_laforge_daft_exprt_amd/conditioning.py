"""On-device batch conditioning.

``DynamicSpeakerStatsManager.process_batch`` (reference src/daft_exprt/dynamic_stats.py:131-195) z-normalises frame- and
symbol-level energy/pitch per speaker with the current support-set statistics, preserving exact zeros, and replaces the
speaker embeddings with the support-set mean -- as a host loop over ``torch.unique(speaker_ids)`` with one ``.item()``
sync and eight boolean-mask gathers per speaker.  Here the statistics live in two small device tables and the whole batch
is conditioned by five launches without a host sync.  Choosing / refreshing the support sets (file I/O,
``refresh_stats``, :60-129) stays with the caller: pass its ``current_stats`` dict.
"""
from __future__ import annotations

import torch

from ._lib import lib
from .ops import _p, _stream


class BatchConditioner:
    def __init__(self, current_stats: dict, device, emb_dim: int = 192):
        """``current_stats``: {speaker_id: {'pitch': {'mean','std'}, 'energy': {'mean','std'}, 'spk_emb': (emb_dim,) tensor}}"""
        self.device = torch.device(device)
        self.update(current_stats, emb_dim)

    def update(self, current_stats: dict, emb_dim: int = 192):
        n = (max(current_stats) + 1) if current_stats else 1
        table = torch.zeros(n, 4)
        table[:, 1] = 1.0
        table[:, 3] = 1.0
        valid = torch.zeros(n, dtype=torch.int32)
        emb = torch.zeros(n, emb_dim)
        for sid, st in current_stats.items():
            table[sid] = torch.tensor([st['energy']['mean'], st['energy']['std'], st['pitch']['mean'], st['pitch']['std']])
            emb[sid] = st['spk_emb'].float()
            valid[sid] = 1
        self.n, self.emb_dim = n, emb_dim
        self.table, self.valid, self.emb = table.to(self.device), valid.to(self.device), emb.to(self.device)

    def _norm(self, x, speaker_ids, which):
        x = x.contiguous()
        out = torch.empty_like(x)
        lib().dx_condition_prosody(_p(x), _p(out), _p(speaker_ids), _p(self.table), _p(self.valid), which, x.shape[0], x.shape[1], self.n, _stream())
        return out

    def process_batch(self, inputs, device=None):
        """Same contract as the reference's ``process_batch(inputs, device)``: 12-tuple in, 12-tuple out."""
        (symbols, durations_float, durations_int, symbols_energy, symbols_pitch, input_lengths,
         frames_energy, frames_pitch, mel_specs, output_lengths, speaker_ids, spk_embs) = inputs
        if not symbols_energy.is_cuda:
            raise RuntimeError('BatchConditioner (MI355X build) runs on the GPU only; there is no CPU path')
        speaker_ids = speaker_ids.contiguous()
        avg = torch.empty(speaker_ids.shape[0], self.emb_dim, dtype=torch.float32, device=speaker_ids.device)
        lib().dx_gather_speaker_rows(_p(self.emb), _p(speaker_ids), _p(self.valid), _p(avg), speaker_ids.shape[0], self.emb_dim, self.n, _stream())
        return (symbols, durations_float, durations_int, self._norm(symbols_energy, speaker_ids, 0), self._norm(symbols_pitch, speaker_ids, 1),
                input_lengths, self._norm(frames_energy, speaker_ids, 0), self._norm(frames_pitch, speaker_ids, 1), mel_specs, output_lengths,
                speaker_ids, avg)
