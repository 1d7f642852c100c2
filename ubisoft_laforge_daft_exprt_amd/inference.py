"""hipGraph-captured inference forward (BASELINE.json config 4).

``DaftExprt.inference`` (reference model.py:1026-1114) interleaves host work (``get_int_durations``: double-precision
Python, ``.item()`` loops; ``int(torch.max(...))``; asserts) with device work, which blocks graph capture.
Here the host part runs FIRST (durations -> integer frames, lengths, T_max, all bit-exact and on the CPU), then the whole
device part -- speaker projection, FiLM generation, phoneme encoder, Gaussian upsampler, frame decoder -- is one
captured HIP graph per (B, L_max, T_max) shape, replayed with new inputs copied into its static buffers.
Lengths are device inputs of the graph, so batches with different per-utterance lengths but the same padded shape
replay the same graph.
"""
from __future__ import annotations

import torch

from . import ops
from .functional import Lengths


class GraphedSynthesizer:
    def __init__(self, model, hparams):
        self.model = model.eval()
        self.hparams = hparams
        self.graphs = {}

    # -- host side: exactly the reference's pre-processing, model.py:1068-1087 --------------------------------------------
    def prepare(self, inputs, pitch_transform, external_prosody):
        symbols, dur_factors, energy_factors, pitch_factors, input_lengths, speaker_ids = inputs
        m = self.model
        duration_preds = external_prosody['duration_preds'] * dur_factors
        duration_preds, durations_int = m.get_int_durations(duration_preds, self.hparams)
        energy = external_prosody['energy_preds'] * energy_factors
        pitch = external_prosody['pitch_preds']
        energy[durations_int == 0] = 0.0
        pitch[durations_int == 0] = 0.0
        if pitch_transform == 'add':
            pitch = m.pitch_shift(pitch, pitch_factors, self.hparams, speaker_ids)
        elif pitch_transform == 'multiply':
            pitch = m.pitch_multiply(pitch, pitch_factors)
        else:
            raise NotImplementedError
        totals = durations_int.sum(dim=1).tolist()
        out_host = [max(1, t) for t in totals]
        return dict(symbols=symbols, in_lens=input_lengths, in_host=input_lengths.tolist(), dur=duration_preds, dur_int=durations_int,
                    energy=energy, pitch=pitch, out_host=out_host, n_frames=max(totals))

    # -- device side -----------------------------------------------------------------------------------------------------
    def _device_forward(self, st):
        m = self.model
        spk = m.spk_projection(ops.l2_normalize(st['spk_embs']), need_dx=False)
        film = m.style_adapter(st['accent_emb'] + spk)
        in_lens = Lengths(st['in_lens'], host=st['in_host'])
        in_lens.i32 = st['in_lens_i32']
        enc = m.phoneme_encoder(st['symbols'], film['phoneme_encoder'], in_lens)
        x, weights = m.gaussian_upsampling(enc, st['dur'], st['dur_int'], st['energy'], st['pitch'], in_lens, n_frames=st['n_frames'])
        out_lens = Lengths(st['out_lens'], host=st['out_host'])
        out_lens.i32 = st['out_lens_i32']
        mel = m.frame_decoder(x, film['frame_decoder'], out_lens)
        return mel, weights

    def __call__(self, inputs, pitch_transform, external_prosody, external_embeddings, external_accent_emb, use_graph=True):
        """Same arguments as ``DaftExprt.inference``; returns the same triple."""
        prep = self.prepare(inputs, pitch_transform, external_prosody)
        dev = prep['symbols'].device
        out_lens = torch.tensor(prep['out_host'], dtype=torch.long, device=dev)
        live = dict(symbols=prep['symbols'].contiguous(), dur=prep['dur'].contiguous(), dur_int=prep['dur_int'].contiguous(),
                    energy=prep['energy'].contiguous(), pitch=prep['pitch'].contiguous(), in_lens=prep['in_lens'],
                    in_lens_i32=prep['in_lens'].to(torch.int32), out_lens=out_lens, out_lens_i32=out_lens.to(torch.int32),
                    spk_embs=external_embeddings.contiguous(), accent_emb=external_accent_emb.contiguous())
        meta = dict(in_host=prep['in_host'], out_host=prep['out_host'], n_frames=prep['n_frames'])
        B, L = prep['symbols'].shape
        key = (B, L, prep['n_frames'], max(prep['out_host']))
        with torch.no_grad():
            if not use_graph:
                mel, weights = self._device_forward({**live, **meta})
            else:
                entry = self.graphs.get(key)
                if entry is None:
                    static = {k: v.clone() for k, v in live.items()}
                    stream = torch.cuda.Stream()
                    stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(stream):                       # warm-up: packs weights, sets kernel attributes
                        self._device_forward({**static, **meta})
                    torch.cuda.current_stream().wait_stream(stream)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        outs = self._device_forward({**static, **meta})
                    entry = (graph, static, outs)
                    self.graphs[key] = entry
                graph, static, outs = entry
                ops.repack_all(self.model.runtime)        # weights changed since capture (load_state_dict, an optimiser step)? one
                                                          # launch rewrites the SAME pack buffers the captured kernels read
                for k, v in live.items():
                    static[k].copy_(v)
                graph.replay()
                mel, weights = outs[0].clone(), outs[1].clone()
        encoder_preds = [prep['dur'], prep['dur_int'], prep['energy'], prep['pitch'], prep['in_lens']]
        return encoder_preds, [mel, out_lens], weights
