"""CPU oracle for the Daft-Exprt acoustic-model hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch, fp32, CPU restatement of the reference algorithm
(`/root/reference/src/daft_exprt/model.py`, `loss.py`, `layers/pitch_predictor.py`,
`extract_features.py:69-125`).  It is the checker the HIP path is compared with
and the timed CPU baseline of ``bench.py``; it is never imported by the product
package (``ubisoft_laforge_daft_exprt_amd``), only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Pinning: the reference ships no tests, fixtures or checkpoints,
so this oracle is pinned against outputs of the reference itself, produced in the
build container by ``tests/golden/make_golden.py`` (imports the reference's own
model/loss on CPU) and committed under ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks every stored tensor.

The op structure of the reference is kept on purpose (slow-path multi-head
attention with a materialised (B*H, N, N) score matrix and head-averaged
weights, broadcast-multiply Gaussian upsampler, per-row positional-encoding
gather), so that the oracle's cost is the reference's cost when it is timed.

The model is written functionally over a ``state_dict`` whose keys and shapes
are the reference checkpoint layout.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def lengths_to_mask(lengths):
    """True where valid.  model.py:14-24."""
    n_max = int(lengths.max())
    return torch.arange(n_max, device=lengths.device)[None, :] < lengths[:, None]


class _ReverseGrad(torch.autograd.Function):
    """Identity forward, -lambda * g backward.  model.py:27-38."""

    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return -ctx.lam * g, None


def linear(sd, prefix, x):
    """LinearNorm, model.py:57-72."""
    return F.linear(x, sd[prefix + '.linear_layer.weight'], sd[prefix + '.linear_layer.bias'])


def conv_cl(sd, prefix, x):
    """ConvNorm1D on channels-last input (B, N, Cin) -> (B, N, Cout); zero 'same' padding.  model.py:75-94."""
    w = sd[prefix + '.conv.weight']
    pad = (w.shape[2] - 1) // 2
    return F.conv1d(x.transpose(1, 2), w, sd[prefix + '.conv.bias'], padding=pad).transpose(1, 2)


_PE_CACHE = {}


def positional_table(dim, max_len=5000, timestep=10000.0):
    """model.py:123-130."""
    key = (dim, max_len)
    if key not in _PE_CACHE:
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, dim, 2).float() * (-math.log(timestep) / dim))
        table = torch.zeros(max_len, dim)
        table[:, 0::2] = torch.sin(pos * div)
        table[:, 1::2] = torch.cos(pos * div)
        _PE_CACHE[key] = table
    return _PE_CACHE[key]


def positional_encoding(lengths, dim):
    """Rows 0..len-1 of the table per batch row, zero beyond.  model.py:132-150 (called with (B,1) lengths)."""
    table = positional_table(dim)
    n_max = int(lengths.max())
    out = torch.zeros(lengths.numel(), n_max, dim)
    for b in range(lengths.numel()):  # per-row gather, as the reference does
        n = int(lengths[b])
        out[b, :n] = table[list(range(n))]
    return out


def multi_head_attention(sd, prefix, x, pad_mask, heads, p_drop, training):
    """nn.MultiheadAttention slow path + dropout + residual LayerNorm.  model.py:153-193.

    x: (B, N, D); pad_mask: (B, N) True at padded keys.
    """
    B, N, D = x.shape
    hd = D // heads
    mha = prefix + '.multi_head_attention'
    qkv = F.linear(x.transpose(0, 1), sd[mha + '.in_proj_weight'], sd[mha + '.in_proj_bias'])  # (N, B, 3D)
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.contiguous().view(N, B * heads, hd).transpose(0, 1) * math.sqrt(1.0 / hd)
    k = k.contiguous().view(N, B * heads, hd).transpose(0, 1)
    v = v.contiguous().view(N, B * heads, hd).transpose(0, 1)
    bias = torch.zeros(B, 1, 1, N).masked_fill(pad_mask[:, None, None, :], float('-inf'))
    bias = bias.expand(B, heads, 1, N).reshape(B * heads, 1, N)
    scores = torch.baddbmm(bias, q, k.transpose(1, 2))  # (B*H, N, N) materialised, as the reference does
    probs = F.softmax(scores, dim=-1)
    probs = F.dropout(probs, p=p_drop, training=training)
    ctx = torch.bmm(probs, v).transpose(0, 1).contiguous().view(N * B, D)
    out = F.linear(ctx, sd[mha + '.out_proj.weight'], sd[mha + '.out_proj.bias']).view(N, B, D).transpose(0, 1)
    _avg_weights = probs.view(B, heads, N, N).mean(dim=1)  # returned and discarded by the caller, model.py:255
    out = F.dropout(out, p=p_drop, training=training)
    return F.layer_norm(out + x, (D,), sd[prefix + '.layer_norm.weight'], sd[prefix + '.layer_norm.bias'], LN_EPS)


def conv_feed_forward(sd, prefix, x, film, p_drop, training):
    """conv k3 -> ReLU -> conv k3 -> dropout -> LN(. + x) -> FiLM.  model.py:196-235."""
    D = x.shape[2]
    h = F.relu(conv_cl(sd, prefix + '.convs.0', x))
    h = F.dropout(conv_cl(sd, prefix + '.convs.2', h), p=p_drop, training=training)
    h = F.layer_norm(h + x, (D,), sd[prefix + '.layer_norm.weight'], sd[prefix + '.layer_norm.bias'], LN_EPS)
    if film is not None:
        h = film[:, None, :D] * h + film[:, None, D:]
    return h


def fft_block(sd, prefix, x, film, pad_mask, cfg, training):
    """model.py:238-259."""
    a = multi_head_attention(sd, prefix + '.attention', x, pad_mask, cfg['attn_nb_heads'], cfg['attn_dropout'], training)
    a = a.masked_fill(pad_mask[:, :, None], 0.0)
    y = conv_feed_forward(sd, prefix + '.feed_forward', a, film, cfg['conv_dropout'], training)
    return y.masked_fill(pad_mask[:, :, None], 0.0)


# ----------------------------------------------------------------------------------------------
# modules
# ----------------------------------------------------------------------------------------------
def _stack_cfg(hp, name):
    cfg = dict(getattr(hp, name))
    cfg.setdefault('hidden_embed_dim', hp.phoneme_encoder['hidden_embed_dim'])
    return cfg


def accent_encoder(sd, frames_energy, frames_pitch, mel, out_lens, hp, training=False):
    """Live AccentEncoder definition, model.py:614-716."""
    cfg = dict(getattr(hp, 'accent_encoder', hp.phoneme_encoder))
    D = cfg['hidden_embed_dim']
    p = 'accent_encoder'
    pos = positional_encoding(out_lens, D)
    energy = conv_cl(sd, p + '.energy_embedding', frames_energy[:, :, None])
    pitch = conv_cl(sd, p + '.pitch_embedding', frames_pitch[:, :, None])
    h = mel.transpose(1, 2)
    for conv_idx, ln_idx in ((0, 2), (4, 6), (8, 10)):
        h = F.relu(conv_cl(sd, f'{p}.convs.{conv_idx}', h))
        h = F.layer_norm(h, (h.shape[2],), sd[f'{p}.convs.{ln_idx}.weight'], sd[f'{p}.convs.{ln_idx}.bias'], LN_EPS)
        h = F.dropout(h, p=cfg['conv_dropout'], training=training)
    pad = ~lengths_to_mask(out_lens)
    h = (h + energy + pitch + pos).masked_fill(pad[:, :, None], 0.0)
    for i in range(cfg['nb_blocks']):
        h = fft_block(sd, f'{p}.blocks.{i}', h, None, pad, cfg, training)
    return h.sum(dim=1) / out_lens[:, None]


def speaker_classifier(sd, accent_emb, hp):
    """GRL -> 3 linears with ReLU.  model.py:809-830."""
    p = 'speaker_classifier.classifier'
    h = _ReverseGrad.apply(accent_emb, hp.lambda_reversal)
    h = F.relu(linear(sd, p + '.1', h))
    h = F.relu(linear(sd, p + '.3', h))
    return linear(sd, p + '.5', h)


def style_adapter(sd, emb, hp):
    """Two linears -> per-module (B, nb_blocks, 2D) FiLM tensors with scalar post-multipliers.  model.py:719-806."""
    D = hp.phoneme_encoder['hidden_embed_dim']
    gammas = linear(sd, 'style_adapter.gammas_predictor', emb)
    betas = linear(sd, 'style_adapter.betas_predictor', emb)
    pm = sd.get('style_adapter.post_multipliers') if getattr(hp, 'post_mult_weight', 0.0) != 0.0 else None
    out, col, blk = {}, 0, 0
    for name in ('phoneme_encoder', 'frame_decoder'):
        nb = getattr(hp, name)['nb_blocks']
        g = gammas[:, col:col + nb * D].reshape(-1, nb, D)
        b = betas[:, col:col + nb * D].reshape(-1, nb, D)
        if pm is not None:
            g = pm[0, blk:blk + nb][None, :, None] * g + 1
            b = pm[1, blk:blk + nb][None, :, None] * b
        else:
            g = 1.0 * g + 1
            b = 1.0 * b
        out[name] = torch.cat((g, b), dim=2)
        col += nb * D
        blk += nb
    return out


def phoneme_encoder(sd, symbols, film, in_lens, hp, training=False):
    """model.py:567-610."""
    cfg = _stack_cfg(hp, 'phoneme_encoder')
    D = cfg['hidden_embed_dim']
    x = F.embedding(symbols, sd['phoneme_encoder.symbols_embedding.weight'])
    pad = ~lengths_to_mask(in_lens)
    x = (x + positional_encoding(in_lens, D)).masked_fill(pad[:, :, None], 0.0)
    for i in range(cfg['nb_blocks']):
        x = fft_block(sd, f'phoneme_encoder.blocks.{i}', x, None if film is None else film[:, i, :], pad, cfg, training)
    return x


def gaussian_upsampling(sd, x, dur_float, dur_int, energy, pitch, in_lens):
    """model.py:417-510 with film_params=None and use_concatenation=False (the only live configuration,
    SURVEY.md §5 'Config / flags')."""
    p = 'gaussian_upsampling'
    d = conv_cl(sd, p + '.duration_projection', dur_float[:, :, None])
    e = conv_cl(sd, p + '.energy_projection', energy[:, :, None])
    f0 = conv_cl(sd, p + '.pitch_projection', pitch[:, :, None])
    x = x + e + f0
    ranges = F.softplus(linear(sd, p + '.projection.0', x + d)).squeeze(2)
    pad = ~lengths_to_mask(in_lens)
    ranges = ranges.masked_fill(pad, 1.0).clamp(min=1e-3)
    means = dur_int.float() / 2
    cumsum = torch.cumsum(dur_int, dim=1)
    means[:, 1:] += cumsum[:, :-1]
    means = torch.nan_to_num(means, nan=0.0, posinf=1e6, neginf=-1e6)[:, :, None]
    stds = torch.nan_to_num(ranges, nan=1.0, posinf=1e6, neginf=1e-3).clamp(min=1e-3)[:, :, None]
    n_frames = int(cumsum.max())
    t = torch.arange(n_frames, dtype=torch.float) + 0.5
    # torch.distributions.Normal.log_prob, written out
    log_prob = -((t - means) ** 2) / (2 * stds ** 2) - stds.log() - math.log(math.sqrt(2 * math.pi))
    probs = torch.exp(log_prob).masked_fill(pad[:, :, None], 0.0)
    weights = probs / (probs.sum(dim=1, keepdim=True) + 1e-20)
    x_up = torch.sum(x.unsqueeze(-1) * weights.unsqueeze(2), dim=1)  # (B, D, T): the reference's B*L*D*T broadcast
    return x_up.permute(0, 2, 1), weights


def frame_decoder(sd, x, film, out_lens, hp, training=False):
    """model.py:513-564."""
    cfg = _stack_cfg(hp, 'frame_decoder')
    D = cfg['hidden_embed_dim']
    pad = ~lengths_to_mask(out_lens)
    x = (x + positional_encoding(out_lens, D)).masked_fill(pad[:, :, None], 0.0)
    for i in range(cfg['nb_blocks']):
        x = fft_block(sd, f'frame_decoder.blocks.{i}', x, film[:, i, :], pad, cfg, training)
    mel = linear(sd, 'frame_decoder.projection', x).masked_fill(pad[:, :, None], 0.0)
    return mel.transpose(1, 2)


def forward(sd, inputs, hp, training=False, external_accent_emb=None, external_spk_emb=None, return_internals=False):
    """DaftExprt.forward, model.py:889-948."""
    if len(inputs) != 12:
        raise ValueError(f'inputs must have 12 elements (including spk_embs). Got {len(inputs)}.')
    (symbols, dur_float, dur_int, sym_energy, sym_pitch, in_lens,
     frm_energy, frm_pitch, mel, out_lens, _speaker_ids, spk_embs) = inputs
    if external_spk_emb is not None:
        spk = external_spk_emb
    else:
        if spk_embs is None:
            raise ValueError('Speaker embeddings (spk_embs) required.')
        spk = linear(sd, 'spk_projection', F.normalize(spk_embs, p=2, dim=-1))
    if external_accent_emb is not None:
        accent = external_accent_emb
    else:
        accent = accent_encoder(sd, frm_energy, frm_pitch, mel, out_lens, hp, training)
    spk_preds = speaker_classifier(sd, accent, hp)
    film = style_adapter(sd, accent + spk, hp)
    enc = phoneme_encoder(sd, symbols, film['phoneme_encoder'], in_lens, hp, training)
    x_up, weights = gaussian_upsampling(sd, enc, dur_float, dur_int, sym_energy, sym_pitch, in_lens)
    mel_pred = frame_decoder(sd, x_up, film['frame_decoder'], out_lens, hp, training)
    pm = sd['style_adapter.post_multipliers'] if getattr(hp, 'post_mult_weight', 0.0) != 0.0 else 1.0
    outputs = (spk_preds, [pm, None, None, film['frame_decoder']],
               [dur_float, sym_energy, sym_pitch, in_lens], [mel_pred, out_lens], weights)
    if return_internals:
        return outputs, dict(accent_emb=accent, spk_emb=spk, film_enc=film['phoneme_encoder'],
                             enc_outputs=enc, x_upsampled=x_up)
    return outputs


# ----------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------
def fold_pitch_predictor(pp_sd):
    """weight_norm (w = g * v / ||v||, norm over dims 1,2) and eval-mode BatchNorm (scale/shift) of the frozen
    PitchPredictor, layers/pitch_predictor.py:27-29, 47-67.  Returns [(w, b, bn_scale, bn_shift)] x4 (bn None on the last)."""
    layers = []
    for conv_idx, bn_idx in ((0, 2), (4, 6), (8, 10), (12, None)):
        v = pp_sd[f'conv_layers.{conv_idx}.conv.weight_v']
        g = pp_sd[f'conv_layers.{conv_idx}.conv.weight_g']
        w = v * (g / v.norm(dim=(1, 2), keepdim=True))
        b = pp_sd[f'conv_layers.{conv_idx}.conv.bias']
        if bn_idx is None:
            layers.append((w, b, None, None))
        else:
            scale = pp_sd[f'conv_layers.{bn_idx}.weight'] / torch.sqrt(pp_sd[f'conv_layers.{bn_idx}.running_var'] + 1e-5)
            shift = pp_sd[f'conv_layers.{bn_idx}.bias'] - pp_sd[f'conv_layers.{bn_idx}.running_mean'] * scale
            layers.append((w, b, scale, shift))
    return layers


def pitch_predictor(pp_sd, mel):
    """(B, 80, T) -> (B, T); eval mode.  layers/pitch_predictor.py:38-74."""
    h = mel
    for w, b, scale, shift in fold_pitch_predictor(pp_sd):
        h = F.conv1d(h, w, b, padding=1)
        if scale is not None:
            h = F.relu(h) * scale[None, :, None] + shift[None, :, None]
    return h.squeeze(1)


def adversarial_weight(iteration, hp):
    """loss.py:52-55."""
    warm, mx = getattr(hp, 'warmup_steps', 10000), getattr(hp, 'adv_max_weight', 1e-2)
    return min(mx, iteration * warm ** -1.5 * mx / warm ** -0.5)


def loss(outputs, targets, iteration, hp, pp_sd=None):
    """DaftExprtLoss.forward, loss.py:57-159.  Returns (total, dict of tensors)."""
    if len(targets) == 8:
        _, _, _, mel_t, out_lens, spk_ids, _frames_energy, frames_pitch = targets
    else:
        _, _, _, mel_t, out_lens, spk_ids = targets
        frames_pitch = None
    spk_preds, film, _, (mel_p, out_lens), _ = outputs
    pm = film[0]
    n_mel = hp.n_mel_channels
    zero = torch.zeros((), dtype=torch.float)
    terms = {}
    if spk_preds is not None:
        ce = F.cross_entropy(spk_preds, spk_ids)
        terms['speaker_ce_raw'] = ce
        terms['speaker_loss'] = adversarial_weight(iteration, hp) * ce
    else:
        terms['speaker_ce_raw'] = zero
        terms['speaker_loss'] = zero
    pmw = getattr(hp, 'post_mult_weight', 1e-3)
    terms['post_mult_loss'] = pmw * (torch.norm(pm, p=2) if (pmw != 0.0 and torch.is_tensor(pm)) else zero)
    msw = getattr(hp, 'mel_spec_weight', 1.0)
    denom = n_mel * out_lens.float()
    terms['mel_spec_l1_loss'] = msw * ((mel_p - mel_t).abs().sum(dim=(1, 2)) / denom).mean()
    terms['mel_spec_l2_loss'] = msw * (((mel_p - mel_t) ** 2).sum(dim=(1, 2)) / denom).mean()
    total = terms['speaker_loss'] + terms['post_mult_loss'] + terms['mel_spec_l1_loss'] + terms['mel_spec_l2_loss']
    ecw = getattr(hp, 'energy_consistency_weight', 0.0)
    terms['energy_consistency_loss'] = zero
    if ecw > 0:
        pe = F.avg_pool1d(torch.norm(torch.exp(mel_p), dim=1)[:, None], 5, 1, 2)[:, 0]
        te = F.avg_pool1d(torch.norm(torch.exp(mel_t), dim=1)[:, None], 5, 1, 2)[:, 0]
        valid = (torch.arange(pe.shape[1])[None, :] < out_lens[:, None]).float()
        terms['energy_consistency_loss'] = (((pe - te) ** 2) * valid).sum() / out_lens.sum().float()
        total = total + ecw * terms['energy_consistency_loss']
    pcw = getattr(hp, 'pitch_consistency_weight', 0.0)
    terms['pitch_consistency_loss'] = zero
    if pcw > 0 and pp_sd is not None and frames_pitch is not None:
        pp = pitch_predictor(pp_sd, mel_p)
        valid = (torch.arange(pp.shape[1])[None, :] < out_lens[:, None]) & (frames_pitch != 0.0)
        terms['pitch_consistency_loss'] = (((pp - frames_pitch) ** 2) * valid.float()).sum() / (valid.float().sum() + 1e-5)
        total = total + pcw * terms['pitch_consistency_loss']
    return total, terms


# ----------------------------------------------------------------------------------------------
# inference-side integer / host math
# ----------------------------------------------------------------------------------------------
def duration_to_integer(float_durations, hp, nb_samples=None):
    """extract_features.py:69-125.  Pure host arithmetic in double precision (Python floats); consumes the list."""
    sr, flt, hop = hp.sampling_rate, hp.filter_length, hp.hop_length
    if nb_samples is None:
        nb_samples = int(sum(e - b for b, e in float_durations) * sr)
    nb_frames = 1 + int((nb_samples - flt) / hop)
    centres = [int(flt / 2) + hop * i for i in range(nb_frames)]
    out, consumed = [], 1
    while consumed <= nb_frames:
        begin, end = float_durations.pop(0)  # IndexError when the frames outlast the phones, as in the reference
        if begin == end:
            raise ValueError
        b, e = int(begin * sr), int(end * sr)
        n = sum(1 for c in centres if b < c <= e)
        out.append(n)
        consumed += n
    if hp.centered:
        edge = int(flt / 2 / hop)
        out[0] += edge
        if float_durations:
            out.append(edge)
        else:
            out[-1] += edge
    else:
        extra = int((flt - hop) / hop)
        left = extra // 2
        out[0] += left
        if float_durations:
            out.append(extra - left)
        else:
            out[-1] += extra - left
    return out


def get_int_durations(duration_preds, hp):
    """model.py:950-973.  Mutates and returns ``duration_preds`` like the reference."""
    dur_min = hp.filter_length / hp.sampling_rate / 2
    duration_preds[duration_preds < dur_min] = 0.0
    dur_int = torch.zeros(duration_preds.shape, dtype=torch.long)
    for b in range(duration_preds.shape[0]):
        end_prev, idx, spans = 0.0, [], []
        for s in range(duration_preds.shape[1]):
            d = duration_preds[b, s].item()
            if d != 0.0:
                idx.append(s)
                spans.append([end_prev, end_prev + d])
                end_prev += d
        dur_int[b, idx] = torch.LongTensor(duration_to_integer(spans, hp))
    return duration_preds, dur_int


def pitch_shift(pitch, factors, hp, speaker_ids):
    """model.py:975-994."""
    unvoiced = pitch == 0.0
    for b in range(pitch.shape[0]):
        st = hp.stats[f'spk {int(speaker_ids[b])}']['pitch']
        hz = torch.exp(st['std'] * pitch[b] + st['mean']) + factors[b]
        pitch[b] = (torch.log(hz) - st['mean']) / st['std']
    pitch[unvoiced] = 0.0
    return pitch


def pitch_multiply(pitch, factors):
    """model.py:996-1024."""
    for b in range(pitch.shape[0]):
        voiced = pitch[b] != 0.0
        mean = pitch[b][voiced].mean()
        dev = (pitch[b] - mean) * factors[b]
        pitch[b] = (pitch[b] + dev).masked_fill(~voiced, 0.0)
    return pitch


def inference(sd, inputs, pitch_transform, hp, external_prosody=None, external_embeddings=None, external_accent_emb=None):
    """DaftExprt.inference, model.py:1026-1114."""
    symbols, dur_factors, energy_factors, pitch_factors, in_lens, speaker_ids = inputs
    if external_embeddings is None:
        raise ValueError('external_embeddings required for inference.')
    spk = linear(sd, 'spk_projection', F.normalize(external_embeddings, p=2, dim=-1))
    if external_accent_emb is None:
        raise ValueError('external_accent_emb required for inference.')
    film = style_adapter(sd, external_accent_emb + spk, hp)
    enc = phoneme_encoder(sd, symbols, film['phoneme_encoder'], in_lens, hp)
    if external_prosody is None:
        raise ValueError('external_prosody must be provided for inference as the internal predictor has been removed.')
    dur = external_prosody['duration_preds'] * dur_factors
    dur, dur_int = get_int_durations(dur, hp)
    energy = external_prosody['energy_preds'] * energy_factors
    pitch = external_prosody['pitch_preds']
    energy[dur_int == 0] = 0.0
    pitch[dur_int == 0] = 0.0
    if pitch_transform == 'add':
        pitch = pitch_shift(pitch, pitch_factors, hp, speaker_ids)
    elif pitch_transform == 'multiply':
        pitch = pitch_multiply(pitch, pitch_factors)
    else:
        raise NotImplementedError
    x_up, weights = gaussian_upsampling(sd, enc, dur, dur_int, energy, pitch, in_lens)
    out_lens = dur_int.sum(dim=1).long()
    out_lens[out_lens == 0] = 1
    assert int(out_lens.max()) == x_up.shape[1]
    mel = frame_decoder(sd, x_up, film['frame_decoder'], out_lens, hp)
    return [dur, dur_int, energy, pitch, in_lens], [mel, out_lens], weights


# ----------------------------------------------------------------------------------------------
# batch conditioning
# ----------------------------------------------------------------------------------------------
def process_batch(inputs, current_stats):
    """DynamicSpeakerStatsManager.process_batch, dynamic_stats.py:131-195 (host loop over the speakers of the batch)."""
    (symbols, dur_f, dur_i, sym_e, sym_p, in_l, frm_e, frm_p, mel, out_l, speaker_ids, spk_embs) = inputs
    frm_e, frm_p, sym_e, sym_p = frm_e.clone(), frm_p.clone(), sym_e.clone(), sym_p.clone()
    dim = next(iter(current_stats.values()))['spk_emb'].shape[0]
    avg = torch.zeros(len(speaker_ids), dim)
    for spk in torch.unique(speaker_ids):
        sid = spk.item()
        if sid not in current_stats:
            continue
        st, rows = current_stats[sid], speaker_ids == spk
        for t, kind in ((frm_e, 'energy'), (sym_e, 'energy'), (frm_p, 'pitch'), (sym_p, 'pitch')):
            v = t[rows]
            zero = v == 0.0
            v = (v - st[kind]['mean']) / st[kind]['std']
            v[zero] = 0.0
            t[rows] = v
        avg[rows] = st['spk_emb']
    return (symbols, dur_f, dur_i, sym_e, sym_p, in_l, frm_e, frm_p, mel, out_l, speaker_ids, avg)
