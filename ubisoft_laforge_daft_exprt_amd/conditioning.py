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
    def __init__(self, current_stats: dict, device, emb_dim: int = 192, capacity: int = 64):
        """``current_stats``: {speaker_id: {'pitch': {'mean','std'}, 'energy': {'mean','std'}, 'spk_emb': (emb_dim,) tensor}}
        ``capacity``: rows the device tables start with (speaker ids 0..capacity-1).

        The three device tables are PERSISTENT: ``update`` (the reference's ``refresh_stats`` every ``stats_refresh_interval``
        iterations, train.py) writes into them in place, because a captured training graph (trainer.Trainer) holds their raw
        pointers and the row count ``n`` as launch arguments.  They are re-allocated only when a speaker id no longer fits; that
        bumps ``generation``, which tells the trainer to drop its captured graphs."""
        self.device = torch.device(device)
        self.emb_dim = int(emb_dim)
        self.n = 0                                 # rows of the tables = the ``S`` argument of both kernels (constant between re-allocations)
        self.generation = 0
        self.table = self.valid = self.emb = None
        self._reserve(max(int(capacity), (max(current_stats) + 1) if current_stats else 1))
        self.update(current_stats, emb_dim)

    def _reserve(self, rows):
        self.n = int(rows)
        self.table = torch.zeros(self.n, 4, dtype=torch.float32, device=self.device)
        self.valid = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.emb = torch.zeros(self.n, self.emb_dim, dtype=torch.float32, device=self.device)
        self.generation += 1

    def update(self, current_stats: dict, emb_dim: int = None):
        """New support-set statistics, written IN PLACE (stream-ordered copies on the current stream: a graph replay enqueued
        afterwards reads the new rows).  Speakers missing from ``current_stats`` become invalid (pass-through), as a fresh table would."""
        if emb_dim is not None and int(emb_dim) != self.emb_dim:
            self.emb_dim = int(emb_dim)
            self._reserve(self.n)
        need = (max(current_stats) + 1) if current_stats else 1
        if need > self.n:
            self._reserve(max(need, 2 * self.n))   # growth: new pointers, new ``n`` -> captured graphs must be dropped (generation)
        table = torch.zeros(self.n, 4)
        table[:, 1] = 1.0
        table[:, 3] = 1.0
        valid = torch.zeros(self.n, dtype=torch.int32)
        emb = torch.zeros(self.n, self.emb_dim)
        for sid, st in current_stats.items():
            table[sid] = torch.tensor([st['energy']['mean'], st['energy']['std'], st['pitch']['mean'], st['pitch']['std']])
            emb[sid] = st['spk_emb'].float()
            valid[sid] = 1
        self.table.copy_(table)
        self.valid.copy_(valid)
        self.emb.copy_(emb)

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
