"""Float -> integer frame durations, host side, double precision (bit-exact requirement).

Reference: ``duration_to_integer`` (src/daft_exprt/extract_features.py:69-125) and ``DaftExprt.get_int_durations``
(src/daft_exprt/model.py:950-973).  The arithmetic is Python ``int()`` truncation of double-precision products, so
it stays on the host; the reference's O(frames x phones) scan is replaced by closed-form counting of the frame
centres ``filter_length/2 + hop*i`` that fall in ``(begin_sample, end_sample]``.
"""
from __future__ import annotations

import torch


def _centres_upto(sample: int, first: int, hop: int, nb_frames: int) -> int:
    """Number of frame centres first + hop*i (0 <= i < nb_frames) that are <= sample."""
    if nb_frames <= 0 or sample < first:
        return 0
    return min(nb_frames, (sample - first) // hop + 1)


def duration_to_integer(float_durations, hparams, nb_samples=None):
    """``float_durations``: list of [begin, end] seconds (consumed from the front, as in the reference)."""
    sr, flt, hop = hparams.sampling_rate, hparams.filter_length, hparams.hop_length
    if nb_samples is None:
        nb_samples = int(sum(end - begin for begin, end in float_durations) * sr)
    nb_frames = 1 + int((nb_samples - flt) / hop)
    first = int(flt / 2)
    out, consumed = [], 1
    while consumed <= nb_frames:
        begin, end = float_durations.pop(0)  # IndexError once the phones run out, like the reference
        if begin == end:
            raise ValueError
        b, e = int(begin * sr), int(end * sr)
        n = _centres_upto(e, first, hop, nb_frames) - _centres_upto(b, first, hop, nb_frames)
        out.append(n)
        consumed += n
    if hparams.centered:
        edge = int(flt / 2 / hop)
        out[0] += edge
        if len(float_durations) != 0:
            out.append(edge)
        else:
            out[-1] += edge
    else:
        extra = int((flt - hop) / hop)
        left = extra // 2
        out[0] += left
        if len(float_durations) != 0:
            out.append(extra - left)
        else:
            out[-1] += extra - left
    return out


def _row_to_integer(row, hparams):
    """One utterance through the Python restatement: (indices of the non-zero symbols, integer durations)."""
    end_prev, idx, spans = 0.0, [], []
    for s, d in enumerate(row):
        if d != 0.0:
            idx.append(s)
            spans.append([end_prev, end_prev + d])
            end_prev += d
    return idx, duration_to_integer(spans, hparams)


def get_int_durations(duration_preds: torch.Tensor, hparams, return_totals=False):
    """Zeroes durations under half an FFT window, converts each row; returns (duration_preds, durations_int on the same device)
    [+ the per-utterance frame totals as Python ints: saves the caller a device reduction and a sync].

    The conversion itself is ``dx_int_durations`` of the host library (C, double precision, the same operations in the same order: one
    pass over the batch instead of a Python loop per symbol).  A row the reference would raise on is re-run through the Python
    restatement above, which raises the reference's exception."""
    from ._lib import lib
    dur_min = hparams.filter_length / hparams.sampling_rate / 2
    duration_preds[duration_preds < dur_min] = 0.0
    host = duration_preds.detach().to(dtype=torch.float32, device='cpu').contiguous()     # the one D2H copy (+ sync) of the conversion
    B, L = host.shape
    out = torch.empty(B, L, dtype=torch.long)
    totals = torch.empty(B, dtype=torch.long)
    status = torch.empty(B, dtype=torch.int32)
    lib().dx_int_durations(host.data_ptr(), B, L, int(hparams.sampling_rate), int(hparams.filter_length), int(hparams.hop_length),
                           int(bool(hparams.centered)), out.data_ptr(), totals.data_ptr(), status.data_ptr())
    if bool(status.any()):
        for b in torch.nonzero(status).flatten().tolist():
            idx, ints = _row_to_integer(host[b].tolist(), hparams)       # raises IndexError / ValueError like the reference ...
            out[b, idx] = torch.tensor(ints, dtype=torch.long)           # ... or RuntimeError (shape mismatch) here
            totals[b] = out[b].sum()
    out_dev = out.to(duration_preds.device, non_blocking=True)
    return (duration_preds, out_dev, totals.tolist()) if return_totals else (duration_preds, out_dev)
