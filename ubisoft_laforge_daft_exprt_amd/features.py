"""Feature-file reader and collator for the reference's on-disk training format.

Format (reference src/daft_exprt/data_loader.py:123-178, extract_features.py:489-539, create_sets.py:46):
  list file      one ``features_dir|feature_file|speaker_id`` line per utterance
  <f>.npy        log-mel (n_mel, T) float32
  <f>.markers    tab-separated ``begin  end  int_dur  symbol  word_begin  word_end`` per symbol
  <f>.symbols_nrg / .symbols_f0 / .frames_nrg / .frames_f0    one float per line
  <f>.spk_emb.npy  (192,) ECAPA embedding
Output: the 14-tuple of ``DaftExprtDataCollate`` (data_loader.py:207-287), in pinned host memory so that
``DaftExprt.parse_batch`` can issue asynchronous H2D copies.  Host-side code, like the reference's.
"""
from __future__ import annotations

import os

import numpy as np
import torch

# vocabulary of the reference for English (symbols.py:16-36): pad, eos, whitespace, punctuation, stressed ARPAbet
_ARPABET_VOWELS = ['AA', 'AE', 'AH', 'AO', 'AW', 'AY', 'EH', 'ER', 'EY', 'IH', 'IY', 'OW', 'OY', 'UH', 'UW']
_ARPABET_ORDER = ['AA', 'AE', 'AH', 'AO', 'AW', 'AY', 'B', 'CH', 'D', 'DH', 'EH', 'ER', 'EY', 'F', 'G', 'HH', 'IH', 'IY', 'JH', 'K', 'L',
                  'M', 'N', 'NG', 'OW', 'OY', 'P', 'R', 'S', 'SH', 'T', 'TH', 'UH', 'UW', 'V', 'W', 'Y', 'Z', 'ZH']
SYMBOLS_ENGLISH = list('_~ ,.!?') + [p + s for p in _ARPABET_ORDER for s in (('0', '1', '2') if p in _ARPABET_VOWELS else ('',))]
assert len(SYMBOLS_ENGLISH) == 76 and SYMBOLS_ENGLISH[0] == '_'


def _floats(path):
    with open(path, 'r', encoding='utf-8') as f:
        return np.array([float(line.strip()) for line in f.readlines()])


def _standardise(values, stats):
    """double-precision z-normalisation that keeps exact zeros (data_loader.py:87-91, :112-116)"""
    zero = np.where(values == 0.)[0]
    values = values - stats['mean']
    values = values / stats['std']
    values[zero] = 0.
    return values


def read_utterance(features_dir, feature_file, speaker_id, hparams, return_raw_stats=False):
    """One utterance -> the 12-tuple of ``DaftExprtDataLoader.get_data`` (data_loader.py:123-178); raises ValueError on the
    same consistency violations."""
    base = os.path.join(features_dir, feature_file)
    mel = torch.from_numpy(np.load(base + '.npy'))
    assert mel.size(0) == hparams.n_mel_channels
    table = list(getattr(hparams, 'symbols', None) or SYMBOLS_ENGLISH)
    symbols, dur_f, dur_i = [], [], []
    with open(base + '.markers', 'r', encoding='utf-8') as f:
        for line in f.readlines():
            begin, end, int_dur, symbol, _, _ = line.strip().split(sep='\t')
            symbols.append(table.index(symbol))
            dur_f.append(float(end) - float(begin))
            dur_i.append(int(int_dur))
    symbols, dur_f, dur_i = torch.IntTensor(symbols), torch.FloatTensor(dur_f), torch.IntTensor(dur_i)
    stats = None if return_raw_stats else hparams.stats[f'spk {speaker_id}']
    sym_e, sym_p = _floats(base + '.symbols_nrg'), _floats(base + '.symbols_f0')
    if stats is not None:
        sym_e, sym_p = _standardise(sym_e, stats['energy']), _standardise(sym_p, stats['pitch'])
    sym_e, sym_p = torch.FloatTensor(sym_e), torch.FloatTensor(sym_p)
    frm_e, frm_p = torch.FloatTensor(_floats(base + '.frames_nrg')), torch.FloatTensor(_floats(base + '.frames_f0'))
    for name, got, want in (('symbols_energy vs symbols length', len(sym_e), len(symbols)), ('symbols_pitch vs symbols length', len(sym_p), len(symbols)),
                            ('frames_energy vs mel frames', len(frm_e), mel.size(1)), ('frames_pitch vs mel frames', len(frm_p), mel.size(1)),
                            ('durations_int sum vs mel frames', int(torch.sum(dur_i)), mel.size(1))):
        if got != want:
            raise ValueError(f'{name} mismatch ({got} vs {want}) for {base}')
    emb_path = base + '.spk_emb.npy'
    emb = torch.from_numpy(np.load(emb_path)).float() if os.path.isfile(emb_path) else None
    return symbols, dur_f, dur_i, sym_e, sym_p, frm_e, frm_p, mel, int(speaker_id), features_dir, feature_file, emb


def collate(batch, hparams, pin_memory=True):
    """``DaftExprtDataCollate.__call__`` (data_loader.py:207-287): sort by symbol length (descending), right zero-pad."""
    n = len(batch)
    input_lengths, order = torch.sort(torch.LongTensor([len(x[0]) for x in batch]), dim=0, descending=True)
    L, T = int(input_lengths[0]), max(x[7].size(1) for x in batch)
    embs = [batch[i][11] for i in order]
    if all(e is None for e in embs):
        raise ValueError('Speaker embeddings (spk_embs) required for every sample. Run training.py pre_process to compute ECAPA '
                         'embeddings and ensure .spk_emb.npy files exist in the feature directories.')
    if any(e is None for e in embs):
        raise ValueError('Mixed presence of speaker embeddings in batch. All samples must have .spk_emb.npy.')
    pin = pin_memory and torch.cuda.is_available()
    z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, pin_memory=pin)
    symbols, dur_i, speaker_ids, out_lens = z(n, L, dtype=torch.long), z(n, L, dtype=torch.long), z(n, dtype=torch.long), z(n, dtype=torch.long)
    dur_f, sym_e, sym_p = z(n, L, dtype=torch.float32), z(n, L, dtype=torch.float32), z(n, L, dtype=torch.float32)
    frm_e, frm_p = z(n, T, dtype=torch.float32), z(n, T, dtype=torch.float32)
    mels = z(n, hparams.n_mel_channels, T, dtype=torch.float32)
    spk_embs = z(n, embs[0].numel(), dtype=torch.float32)
    dirs, files = [], []
    for i, src in enumerate(order.tolist()):
        s, df, di, se, sp, fe, fp, mel, sid, d, f, emb = batch[src]
        symbols[i, :s.size(0)] = s
        dur_f[i, :df.size(0)] = df
        dur_i[i, :di.size(0)] = di
        sym_e[i, :se.size(0)] = se
        sym_p[i, :sp.size(0)] = sp
        frm_e[i, :fe.size(0)] = fe
        frm_p[i, :fp.size(0)] = fp
        mels[i, :, :mel.size(1)] = mel
        out_lens[i] = mel.size(1)
        speaker_ids[i] = sid
        spk_embs[i] = emb
        dirs.append(d)
        files.append(f)
    return symbols, dur_f, dur_i, sym_e, sym_p, input_lengths, frm_e, frm_p, mels, out_lens, speaker_ids, dirs, files, spk_embs


class FeatureSet:
    """Utterance list sharded by rank with ``DistributedSampler(shuffle=False)`` semantics (data_loader.py:310): the list is
    first padded BY REPETITION (wrapping around to its head) to a multiple of ``world``, then utterance i goes to rank i mod world,
    so every rank owns the same number of utterances -- and therefore runs the same number of steps: a rank with one more batch
    than its peers would block forever in the gradient all-reduce."""

    def __init__(self, list_file, hparams, batch_size, rank=0, world=1, return_raw_stats=False, drop_last=True):
        with open(list_file, 'r', encoding='utf-8') as f:
            rows = [line.strip().split('|') for line in f if line.strip()]
        if world > 1 and rows and len(rows) % world:
            pad = world - len(rows) % world
            rows = rows + (rows * ((pad + len(rows) - 1) // len(rows)))[:pad]
        self.rows = rows[rank::world]
        self.hparams, self.batch_size, self.raw, self.drop_last = hparams, batch_size, return_raw_stats, drop_last

    def __len__(self):
        n = len(self.rows) // self.batch_size
        return n if self.drop_last or len(self.rows) % self.batch_size == 0 else n + 1

    def __iter__(self):
        for i in range(0, len(self.rows), self.batch_size):
            rows = self.rows[i:i + self.batch_size]
            if len(rows) < self.batch_size and self.drop_last:
                return
            yield collate([read_utterance(d, f, int(s), self.hparams, self.raw) for d, f, s in rows], self.hparams)
