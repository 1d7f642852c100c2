"""``DaftExprt`` with the reference's module surface and checkpoint layout, computed by the gfx950 kernels.

Drop-in contract: constructor ``DaftExprt(hparams, is_training=True)``, ``parse_batch``, ``forward``,
``inference``, ``get_int_durations``, ``pitch_shift``, ``pitch_multiply`` and a ``state_dict()`` whose 184 keys and shapes
are those of src/daft_exprt/model.py (live class definitions, :513-856).  The sub-modules below only OWN parameters under
the reference's names; their ``forward`` methods call ``functional.*Fn`` (fixed sequences of HIP kernel launches).
There is no PyTorch fallback: on a machine without the built extension or without a GPU the forward raises.
"""
from __future__ import annotations

import math

import os

import torch
from torch import nn

from . import functional as Fx
from . import ops
from .durations import get_int_durations as _get_int_durations
from .functional import Lengths


def _xavier_(t, gain_name='linear'):
    nn.init.xavier_uniform_(t, gain=nn.init.calculate_gain(gain_name))
    return t


class _Affine(nn.Module):
    """weight/bias holder (Conv1d, Linear, LayerNorm parameters keep the reference's key names)."""

    def __init__(self, wshape, bias_len, gain_name=None, layer_norm=False):
        super().__init__()
        if layer_norm:
            self.weight = nn.Parameter(torch.ones(wshape))
            self.bias = nn.Parameter(torch.zeros(bias_len))
        else:
            w = torch.empty(wshape)
            _xavier_(w, gain_name or 'linear')
            fan_in = w[0].numel()
            bound = 1.0 / math.sqrt(fan_in)
            self.weight = nn.Parameter(w)
            self.bias = nn.Parameter(torch.empty(bias_len).uniform_(-bound, bound))
        self._pack = None

    @property
    def pack(self):
        rt = getattr(self, '_dx_rt', None) or ops.DEFAULT
        if self._pack is None or self._pack.weight is not self.weight or self._pack.rt is not rt:
            self._pack = ops.PackedWeight(self.weight, rt)
        return self._pack


class ConvNorm1D(nn.Module):
    """Parameters of a channels-last Conv1d (reference model.py:75-94): key prefix ``conv.``"""

    def __init__(self, cin, cout, kernel_size, w_init_gain='linear'):
        super().__init__()
        self.conv = _Affine((cout, cin, kernel_size), cout, w_init_gain)


class LinearNorm(nn.Module):
    """Parameters of a Linear layer (reference model.py:57-72): key prefix ``linear_layer.``"""

    def __init__(self, cin, cout, w_init_gain='linear'):
        super().__init__()
        self.linear_layer = _Affine((cout, cin), cout, w_init_gain)

    def forward(self, x, relu=False, grad_scale=1.0, need_dx=True):
        p = self.linear_layer
        return Fx.LinearFn.apply(x, p.weight, p.bias, p.pack, relu, grad_scale, None, need_dx)


class _MHAParams(nn.Module):
    """Parameter names of nn.MultiheadAttention (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias)."""

    def __init__(self, dim):
        super().__init__()
        self.in_proj_weight = nn.Parameter(_xavier_(torch.empty(3 * dim, dim)))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = _Affine((dim, dim), dim)
        nn.init.zeros_(self.out_proj.bias)
        self._pack = None

    @property
    def in_pack(self):
        rt = getattr(self, '_dx_rt', None) or ops.DEFAULT
        if self._pack is None or self._pack.weight is not self.in_proj_weight or self._pack.rt is not rt:
            self._pack = ops.PackedWeight(self.in_proj_weight, rt)
        return self._pack


class MultiHeadAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.multi_head_attention = _MHAParams(cfg['hidden_embed_dim'])
        self.layer_norm = _Affine(cfg['hidden_embed_dim'], cfg['hidden_embed_dim'], layer_norm=True)


class PositionWiseConvFF(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D, Fc, k = cfg['hidden_embed_dim'], cfg['conv_channels'], cfg['conv_kernel']
        self.convs = nn.ModuleList([ConvNorm1D(D, Fc, k, 'relu'), nn.Identity(), ConvNorm1D(Fc, D, k, 'linear'), nn.Identity()])
        self.layer_norm = _Affine(D, D, layer_norm=True)


class FFTBlock(nn.Module):
    """reference model.py:238-259"""

    def __init__(self, cfg):
        super().__init__()
        if cfg['conv_kernel'] != 3 or cfg['hidden_embed_dim'] != 128 or cfg['hidden_embed_dim'] // cfg['attn_nb_heads'] != 64:
            raise NotImplementedError('the gfx950 kernels are built for conv_kernel=3, hidden_embed_dim=128, head_dim=64 '
                                      f'(reference defaults, hparams.py:106-127); got {cfg}')
        self.cfg = dict(cfg)
        self.attention = MultiHeadAttention(cfg)
        self.feed_forward = PositionWiseConvFF(cfg)

    def forward(self, x, film_params, lens: Lengths, qkv_pre=None, next_block=None):
        """-> (y, qkv_next): ``qkv_pre`` is this block's q/k/v when the previous block's last launch produced it; ``qkv_next`` the next
        block's, when this block's fused feed-forward launch could produce it (else None)."""
        mha, ln1 = self.attention.multi_head_attention, self.attention.layer_norm
        c1, c2, ln2 = self.feed_forward.convs[0].conv, self.feed_forward.convs[2].conv, self.feed_forward.layer_norm
        packs = {'in': mha.in_pack, 'out': mha.out_proj.pack, 'c1': c1.pack, 'c2': c2.pack,
                 'params': {'in_w': mha.in_proj_weight, 'in_b': mha.in_proj_bias, 'out_w': mha.out_proj.weight, 'out_b': mha.out_proj.bias,
                            'ln1_w': ln1.weight, 'ln1_b': ln1.bias, 'c1_w': c1.weight, 'c1_b': c1.bias, 'c2_w': c2.weight, 'c2_b': c2.bias,
                            'ln2_w': ln2.weight, 'ln2_b': ln2.bias}}
        nxt = None
        if next_block is not None:
            nm = next_block.attention.multi_head_attention
            nxt = (nm.in_pack, nm.in_proj_bias)
        return Fx.FFTBlockFn.apply(x, film_params, lens, packs, self.cfg, self.training,
                                   mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                                   ln1.weight, ln1.bias, c1.weight, c1.bias, c2.weight, c2.bias, ln2.weight, ln2.bias, qkv_pre, nxt)


def _run_blocks(blocks, x, film, lens):
    """The FFT blocks of one stack; a block's last launch hands the next block its q/k/v where it can (FFTBlock.forward)."""
    qkv = None
    for i, block in enumerate(blocks):
        x, qkv = block(x, None if film is None else film[i], lens, qkv_pre=qkv, next_block=blocks[i + 1] if i + 1 < len(blocks) else None)
    return x


_PE_TABLES = {}


def positional_table(dim, device, max_len=5000, timestep=10000.0):
    """Sinusoid table of reference model.py:123-130 (not a parameter, not in the state dict), resident on the device."""
    key = (dim, str(device))
    if key not in _PE_TABLES:
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(timestep) / dim))
        table = torch.zeros(max_len, dim)
        table[:, 0::2] = torch.sin(pos * div)
        table[:, 1::2] = torch.cos(pos * div)
        _PE_TABLES[key] = table.to(device)
    return _PE_TABLES[key]


def _stack_cfg(hparams, name):
    cfg = dict(getattr(hparams, name))
    cfg.setdefault('hidden_embed_dim', hparams.phoneme_encoder['hidden_embed_dim'])
    return cfg


class AccentEncoder(nn.Module):
    """reference model.py:614-716 (live definition)"""

    def __init__(self, hparams):
        super().__init__()
        cfg = dict(getattr(hparams, 'accent_encoder', hparams.phoneme_encoder))
        D, Fc, k = cfg['hidden_embed_dim'], cfg['conv_channels'], cfg['conv_kernel']
        if Fc != 1024:
            raise NotImplementedError('prenet LayerNorm kernels are built for conv_channels=1024')
        self.cfg = cfg
        self.energy_embedding = ConvNorm1D(1, D, k)
        self.pitch_embedding = ConvNorm1D(1, D, k)
        ident = nn.Identity
        self.convs = nn.ModuleList([
            ConvNorm1D(hparams.n_mel_channels, Fc, k, 'relu'), ident(), _Affine(Fc, Fc, layer_norm=True), ident(),
            ConvNorm1D(Fc, Fc, k, 'relu'), ident(), _Affine(Fc, Fc, layer_norm=True), ident(),
            ConvNorm1D(Fc, D, k, 'relu'), ident(), _Affine(D, D, layer_norm=True), ident()])
        self.blocks = nn.ModuleList([FFTBlock(cfg) for _ in range(cfg['nb_blocks'])])

    def forward(self, frames_energy, frames_pitch, mel_specs, output_lengths):
        lens = output_lengths if isinstance(output_lengths, Lengths) else Lengths(output_lengths)
        c = self.convs
        packs = {'p0': c[0].conv.pack, 'p1': c[4].conv.pack, 'p2': c[8].conv.pack,
                 'params': {'c0_w': c[0].conv.weight, 'c0_b': c[0].conv.bias, 'l0_w': c[2].weight, 'l0_b': c[2].bias,
                            'c1_w': c[4].conv.weight, 'c1_b': c[4].conv.bias, 'l1_w': c[6].weight, 'l1_b': c[6].bias,
                            'c2_w': c[8].conv.weight, 'c2_b': c[8].conv.bias, 'l2_w': c[10].weight, 'l2_b': c[10].bias}}
        pe = positional_table(self.cfg['hidden_embed_dim'], mel_specs.device)
        rt = getattr(self, '_dx_rt', None)
        p_drop = self.cfg['conv_dropout']
        y0 = Fx.AccentFront0Fn.apply(mel_specs, lens, packs, p_drop, self.training, c[0].conv.weight, c[0].conv.bias, c[2].weight, c[2].bias)
        y0 = Fx.cut(rt, 3, y0)                     # phase D of a trainer's backward: prenet layer 0 (its 1 MB of gradients is all that is exchanged last)
        x = Fx.AccentFront12Fn.apply(y0, frames_energy, frames_pitch, lens, packs, pe, p_drop, self.training,
                                     c[4].conv.weight, c[4].conv.bias, c[6].weight, c[6].bias,
                                     c[8].conv.weight, c[8].conv.bias, c[10].weight, c[10].bias,
                                     self.energy_embedding.conv.weight, self.energy_embedding.conv.bias,
                                     self.pitch_embedding.conv.weight, self.pitch_embedding.conv.bias)
        x = Fx.cut(rt, 2, x)                       # phase C: prenet layers 2, 1 + the prosody embeddings; phase B: the four FFT blocks
        x = _run_blocks(self.blocks, x, None, lens)
        return Fx.MeanPoolFn.apply(x, lens, getattr(self, '_dx_rt', None))


class SpeakerClassifier(nn.Module):
    """reference model.py:809-830: gradient reversal (-lambda on the way back) + 3 linears with ReLU"""

    def __init__(self, hparams):
        super().__init__()
        D = hparams.phoneme_encoder['hidden_embed_dim']
        self.lambda_ = hparams.lambda_reversal
        self.classifier = nn.ModuleList([nn.Identity(), LinearNorm(D, D, 'relu'), nn.Identity(), LinearNorm(D, D, 'relu'),
                                         nn.Identity(), LinearNorm(D, hparams.n_speakers, 'linear')])
        self._padded = None

    def forward(self, x):
        h = self.classifier[1](x, relu=True, grad_scale=-float(self.lambda_))   # GRL folded into the first input gradient
        h = self.classifier[3](h, relu=True)
        last = self.classifier[5].linear_layer
        rt = getattr(self, '_dx_rt', None) or ops.DEFAULT
        if self._padded is None or self._padded.weight is not last.weight or self._padded.rt is not rt:
            self._padded = Fx.PaddedLinear(last.weight, last.bias, rt)
        return Fx.PaddedLinearFn.apply(h, last.weight, last.bias, self._padded)


class FilmSet(list):
    """Per-block FiLM tensors (B, 2C) of one module, plus the (B, nb, 2C) tensor the reference returns for it."""

    def __init__(self, blocks, tensor):
        super().__init__(blocks)
        self.tensor = tensor


def _film_blocks(film_params):
    if film_params is None or isinstance(film_params, FilmSet):
        return film_params
    return Fx.SplitFilmFn.apply(film_params)


class StyleAdapter(nn.Module):
    """reference model.py:719-806"""

    def __init__(self, hparams):
        super().__init__()
        D = hparams.phoneme_encoder['hidden_embed_dim']
        hidden = getattr(hparams, 'accent_encoder', hparams.phoneme_encoder)['hidden_embed_dim']
        self.module_params = {'phoneme_encoder': (hparams.phoneme_encoder['nb_blocks'], D),
                              'frame_decoder': (hparams.frame_decoder['nb_blocks'], D)}
        total = sum(nb * ch for nb, ch in self.module_params.values())
        self.gammas_predictor = LinearNorm(hidden, total)
        self.betas_predictor = LinearNorm(hidden, total)
        if getattr(hparams, 'post_mult_weight', 0.0) != 0.0:
            self.post_mult_weight = hparams.post_mult_weight
            self.post_multipliers = nn.Parameter(_xavier_(torch.empty(2, sum(nb for nb, _ in self.module_params.values()))))
        else:
            self.post_mult_weight = 0.0
            self.post_multipliers = 1.0

    def forward(self, style_embedding):
        gammas = self.gammas_predictor(style_embedding)
        betas = self.betas_predictor(style_embedding)
        out, col, blk = {}, 0, 0
        B = gammas.shape[0]
        chans = {ch for _, ch in self.module_params.values()}
        if len(chans) == 1:
            # every module has the same width: the scalar post-multiplier affine, the (gamma | beta) concatenation and the split
            # into per-block tensors are done ONCE for all blocks (same element-wise arithmetic, a third of the glue launches)
            ch = chans.pop()
            nb_all = sum(nb for nb, _ in self.module_params.values())
            if gammas.is_cuda and os.environ.get('DX_FILM_FUSED', '1') != '0':      # one launch forward, one backward (functional.FilmAffineFn)
                pm = self.post_multipliers if self.post_mult_weight != 0.0 else None
                *blocks, whole = Fx.FilmAffineFn.apply(gammas, betas, pm, nb_all, getattr(self, '_dx_rt', None))
                for name, (nb, _) in self.module_params.items():
                    # the (B, nb, 2C) tensor of the reference's return value: a strided view of the block-major buffer
                    out[name] = FilmSet(blocks[blk:blk + nb], whole[blk:blk + nb].transpose(0, 1))
                    blk += nb
                return out
            g = gammas.view(B, nb_all, ch)
            b = betas.view(B, nb_all, ch)
            if self.post_mult_weight != 0.0:
                g = self.post_multipliers[0][None, :, None] * g + 1
                b = self.post_multipliers[1][None, :, None] * b
            else:
                g = g + 1
            film_all = torch.cat((g, b), dim=2)
            blocks = Fx.SplitFilmFn.apply(film_all)
            for name, (nb, _) in self.module_params.items():
                out[name] = FilmSet(blocks[blk:blk + nb], film_all[:, blk:blk + nb])
                blk += nb
            return out
        for name, (nb, ch) in self.module_params.items():
            # (B, nb, ch) scalar post-multiplier affine: O(B * 1024) element-wise glue on the autograd tape
            g = gammas[:, col:col + nb * ch].view(B, nb, ch)
            b = betas[:, col:col + nb * ch].view(B, nb, ch)
            if self.post_mult_weight != 0.0:
                g = self.post_multipliers[0, blk:blk + nb][None, :, None] * g + 1
                b = self.post_multipliers[1, blk:blk + nb][None, :, None] * b
            else:
                g = g + 1
            out[name] = torch.cat((g, b), dim=2)
            col += nb * ch
            blk += nb
        return out


class PhonemeEncoder(nn.Module):
    """reference model.py:567-610"""

    def __init__(self, hparams):
        super().__init__()
        cfg = _stack_cfg(hparams, 'phoneme_encoder')
        self.cfg = cfg
        self.symbols_embedding = nn.Embedding(hparams.n_symbols, cfg['hidden_embed_dim'])
        nn.init.xavier_uniform_(self.symbols_embedding.weight.data)
        self.blocks = nn.ModuleList([FFTBlock(cfg) for _ in range(cfg['nb_blocks'])])

    def forward(self, x, film_params, input_lengths):
        lens = input_lengths if isinstance(input_lengths, Lengths) else Lengths(input_lengths)
        pe = positional_table(self.cfg['hidden_embed_dim'], x.device)
        h = Fx.EmbedPosFn.apply(x, self.symbols_embedding.weight, pe, lens, getattr(self, '_dx_rt', None))
        film = _film_blocks(film_params)
        return _run_blocks(self.blocks, h, film, lens)


class GaussianUpsamplingModule(nn.Module):
    """reference model.py:385-510 (film_params=None, use_concatenation=False: the only live configuration)"""

    def __init__(self, hparams):
        super().__init__()
        D = hparams.phoneme_encoder['hidden_embed_dim']
        gum = dict(hparams.gaussian_upsampling_module)
        if gum.get('use_concatenation', False):
            raise NotImplementedError('use_concatenation=True is not a live configuration of the reference (SURVEY.md §5)')
        k = gum['conv_kernel']
        self.duration_projection = ConvNorm1D(1, D, k)
        self.energy_projection = ConvNorm1D(1, D, k)
        self.pitch_projection = ConvNorm1D(1, D, k)
        self.projection = nn.ModuleList([LinearNorm(D, 1, 'relu'), nn.Identity()])

    def forward(self, x, durations_float, durations_int, energies, pitch, input_lengths, film_params=None, n_frames=None):
        if film_params is not None:
            raise NotImplementedError('FiLM on the upsampling projections is never enabled by the reference forward (model.py:930-933)')
        lens = input_lengths if isinstance(input_lengths, Lengths) else Lengths(input_lengths)
        if n_frames is None:
            n_frames = int(durations_int.sum(dim=1).max())      # reference: torch.max(cumsum), model.py:497 (host sync)
        d, e, p, r = self.duration_projection.conv, self.energy_projection.conv, self.pitch_projection.conv, self.projection[0].linear_layer
        return Fx.GaussianUpsampleFn.apply(x, durations_float, durations_int, energies, pitch, lens, n_frames,
                                           getattr(self, '_dx_rt', None), d.weight, d.bias, e.weight, e.bias, p.weight, p.bias, r.weight, r.bias)


class FrameDecoder(nn.Module):
    """reference model.py:513-564"""

    def __init__(self, hparams, is_training=True):
        super().__init__()
        D = getattr(hparams, 'frame_decoder_input_dim', hparams.phoneme_encoder['hidden_embed_dim'])
        hparams.frame_decoder['hidden_embed_dim'] = D          # the reference inserts this key too, model.py:534
        cfg = dict(hparams.frame_decoder)
        self.cfg = cfg
        self.blocks = nn.ModuleList([FFTBlock(cfg) for _ in range(cfg['nb_blocks'])])
        self.projection = LinearNorm(D, hparams.n_mel_channels)

    def forward(self, x, film_params, output_lengths):
        lens = output_lengths if isinstance(output_lengths, Lengths) else Lengths(output_lengths)
        pe = positional_table(self.cfg['hidden_embed_dim'], x.device)
        h = Fx.AddPosFn.apply(x, pe, lens)
        film = _film_blocks(film_params)
        h = _run_blocks(self.blocks, h, film, lens)
        p = self.projection.linear_layer
        return Fx.MelProjectionFn.apply(h, p.weight, p.bias, p.pack, lens)


class DaftExprt(nn.Module):
    """reference model.py:832-1114"""

    def __init__(self, hparams, is_training=True):
        super().__init__()
        self.n_speakers = hparams.n_speakers
        self.hidden_embed_dim = hparams.phoneme_encoder['hidden_embed_dim']
        self.accent_encoder = AccentEncoder(hparams)
        self.speaker_classifier = SpeakerClassifier(hparams)
        self.style_adapter = StyleAdapter(hparams)
        self.phoneme_encoder = PhonemeEncoder(hparams)
        self.gaussian_upsampling = GaussianUpsamplingModule(hparams)
        self.frame_decoder = FrameDecoder(hparams, is_training=is_training)
        self.spk_projection = LinearNorm(getattr(hparams, 'external_emb_dim', 192), self.hidden_embed_dim)
        # execution state of THIS model (operand precision, pack epoch, gradient sink): shared by all of its sub-modules, read
        # by nobody else.  It starts from the package default (``set_precision``) and is switched with ``model.set_precision``.
        self.runtime = ops.Runtime(ops.DEFAULT.precision)
        for m in self.modules():
            m._dx_rt = self.runtime

    def set_precision(self, name: str):
        """'f32' (exact-f32 MFMA operands, parity mode) or 'bf16' (bf16 MFMA operands, fp32 accumulate; throughput mode)."""
        self.runtime.set_precision(name)
        return self

    # -- batch plumbing ------------------------------------------------------------------------------------------------
    def parse_batch(self, device, batch):
        """reference model.py:858-887.  Also remembers the (CPU) lengths so that forward needs no device->host sync."""
        if len(batch) != 14:
            raise ValueError(f'Batch must have 14 elements (including speaker embeddings). Got {len(batch)}. '
                             'Run training.py pre_process to compute ECAPA embeddings and ensure .spk_emb.npy files exist.')
        (symbols, durations_float, durations_int, symbols_energy, symbols_pitch, input_lengths, frames_energy, frames_pitch,
         mel_specs, output_lengths, speaker_ids, _feature_dirs, _feature_files, spk_embs) = batch
        host = {}
        if not input_lengths.is_cuda:
            host['in'] = input_lengths.tolist()
            host['out'] = output_lengths.tolist()
        to = lambda t, dt: t.to(device, non_blocking=True).to(dt)
        spk_embs = to(spk_embs, torch.float32)
        symbols, durations_int = to(symbols, torch.long), to(durations_int, torch.long)
        durations_float, symbols_energy, symbols_pitch = (to(t, torch.float32) for t in (durations_float, symbols_energy, symbols_pitch))
        input_lengths, output_lengths, speaker_ids = (to(t, torch.long) for t in (input_lengths, output_lengths, speaker_ids))
        frames_energy, frames_pitch, mel_specs = (to(t, torch.float32) for t in (frames_energy, frames_pitch, mel_specs))
        if host:                                   # ride along on the tensor objects themselves (no global cache to go stale)
            input_lengths._dx_host_lengths = host['in']
            output_lengths._dx_host_lengths = host['out']
        inputs = (symbols, durations_float, durations_int, symbols_energy, symbols_pitch, input_lengths,
                  frames_energy, frames_pitch, mel_specs, output_lengths, speaker_ids, spk_embs)
        targets = (durations_float, symbols_energy, symbols_pitch, mel_specs, output_lengths, speaker_ids)
        return inputs, targets

    @staticmethod
    def _lengths(t):
        """A fresh ``Lengths`` per forward call (never reused across calls: the tensor's contents may have changed), left on the tensor
        object so that the loss, which is handed the same output-length tensor, does not build a second one (an int64 -> int32 launch)."""
        obj = Lengths(t, host=getattr(t, '_dx_host_lengths', None))
        try:
            t._dx_lengths = obj
        except AttributeError:
            pass
        return obj

    @staticmethod
    def _require_gpu(t):
        if not t.is_cuda:
            raise RuntimeError('DaftExprt (MI355X build) runs on the GPU only: move the model and the batch to a HIP device; '
                               'there is no CPU path')

    # -- training forward ----------------------------------------------------------------------------------------------
    def forward(self, inputs, external_accent_emb=None, external_spk_emb=None):
        """reference model.py:889-948"""
        if len(inputs) != 12:
            raise ValueError(f'inputs must have 12 elements (including spk_embs). Got {len(inputs)}.')
        (symbols, durations_float, durations_int, symbols_energy, symbols_pitch, input_lengths,
         frames_energy, frames_pitch, mel_specs, output_lengths, _speaker_ids, spk_embs) = inputs
        self._require_gpu(symbols)
        in_lens, out_lens = self._lengths(input_lengths), self._lengths(output_lengths)
        if external_spk_emb is not None:
            spk_emb = external_spk_emb
        else:
            if spk_embs is None:
                raise ValueError('Speaker embeddings (spk_embs) required. Precompute ECAPA and provide .spk_emb.npy in data.')
            spk_emb = self.spk_projection(ops.l2_normalize(spk_embs.contiguous()), need_dx=False)
        if external_accent_emb is not None:
            accent_emb = external_accent_emb
        else:
            accent_emb = self.accent_encoder(frames_energy, frames_pitch, mel_specs, out_lens)
            # phased backward (trainer.Trainer): everything downstream of the accent embedding first -- its gradient buckets
            # can then be exchanged while the accent encoder's backward (the largest module) still runs
            accent_emb = Fx.cut(self.runtime, 1, accent_emb)
        speaker_preds = self.speaker_classifier(accent_emb)
        film = self.style_adapter(accent_emb + spk_emb)
        enc_outputs = self.phoneme_encoder(symbols, film['phoneme_encoder'], in_lens)
        x, weights = self.gaussian_upsampling(enc_outputs, durations_float, durations_int, symbols_energy, symbols_pitch,
                                              in_lens, film_params=None, n_frames=out_lens.max)
        mel_preds = self.frame_decoder(x, film['frame_decoder'], out_lens)
        fd = film['frame_decoder']
        film_params = [self.style_adapter.post_multipliers, None, None, fd.tensor if isinstance(fd, FilmSet) else fd]
        encoder_preds = [durations_float, symbols_energy, symbols_pitch, input_lengths]
        decoder_preds = [mel_preds, output_lengths]
        return speaker_preds, film_params, encoder_preds, decoder_preds, weights

    # -- inference -----------------------------------------------------------------------------------------------------
    def get_int_durations(self, duration_preds, hparams):
        """reference model.py:950-973 (host, double precision, bit-exact)"""
        return _get_int_durations(duration_preds, hparams)

    def pitch_shift(self, pitch_preds, pitch_factors, hparams, speaker_ids):
        """reference model.py:975-994: per-speaker Hz-domain shift, unvoiced (== 0) preserved"""
        unvoiced = pitch_preds == 0.0
        ids = speaker_ids.tolist()
        mean = torch.tensor([hparams.stats[f'spk {i}']['pitch']['mean'] for i in ids], device=pitch_preds.device)[:, None]
        std = torch.tensor([hparams.stats[f'spk {i}']['pitch']['std'] for i in ids], device=pitch_preds.device)[:, None]
        hz = torch.exp(std * pitch_preds + mean) + pitch_factors
        pitch_preds.copy_((torch.log(hz) - mean) / std)
        pitch_preds[unvoiced] = 0.0
        return pitch_preds

    def pitch_multiply(self, pitch_preds, pitch_factors):
        """reference model.py:996-1024: scale the deviation from the voiced mean"""
        voiced = pitch_preds != 0.0
        count = voiced.sum(dim=1, keepdim=True)
        mean = (pitch_preds * voiced).sum(dim=1, keepdim=True) / count     # NaN for an all-unvoiced row, like torch.mean of nothing
        out = pitch_preds + (pitch_preds - mean) * pitch_factors
        pitch_preds.copy_(torch.where(voiced, out, torch.zeros_like(out)))
        return pitch_preds

    def inference(self, inputs, pitch_transform, hparams, external_prosody=None, external_embeddings=None, external_accent_emb=None):
        """reference model.py:1026-1114"""
        symbols, dur_factors, energy_factors, pitch_factors, input_lengths, speaker_ids = inputs
        if external_embeddings is None:
            raise ValueError('external_embeddings required for inference. Provide ECAPA speaker embedding.')
        self._require_gpu(symbols)
        spk_emb = self.spk_projection(ops.l2_normalize(external_embeddings.contiguous()), need_dx=False)
        if external_accent_emb is None:
            raise ValueError('external_accent_emb required for inference. Provide --accent_emb_audios_dir or use a checkpoint '
                             'with memorized_accent_emb (e.g. from adapt_accent).')
        film = self.style_adapter(external_accent_emb + spk_emb)
        in_lens = self._lengths(input_lengths)
        enc_outputs = self.phoneme_encoder(symbols, film['phoneme_encoder'], in_lens)
        if external_prosody is None:
            raise ValueError('external_prosody must be provided for inference as the internal predictor has been removed.')
        duration_preds = external_prosody['duration_preds'] * dur_factors
        duration_preds, durations_int = self.get_int_durations(duration_preds, hparams)
        energy_preds = external_prosody['energy_preds'] * energy_factors
        pitch_preds = external_prosody['pitch_preds']
        energy_preds[durations_int == 0] = 0.0
        pitch_preds[durations_int == 0] = 0.0
        if pitch_transform == 'add':
            pitch_preds = self.pitch_shift(pitch_preds, pitch_factors, hparams, speaker_ids)
        elif pitch_transform == 'multiply':
            pitch_preds = self.pitch_multiply(pitch_preds, pitch_factors)
        else:
            raise NotImplementedError
        totals = durations_int.sum(dim=1)
        host_totals = totals.tolist()
        symbols_upsamp, weights = self.gaussian_upsampling(enc_outputs, duration_preds, durations_int, energy_preds, pitch_preds,
                                                           in_lens, film_params=None, n_frames=max(host_totals))
        output_lengths = totals.long()
        output_lengths[output_lengths == 0] = 1
        out_lens = Lengths(output_lengths, host=[max(1, t) for t in host_totals])
        assert out_lens.max == symbols_upsamp.size(1)
        mel_spec_preds = self.frame_decoder(symbols_upsamp, film['frame_decoder'], out_lens)
        encoder_preds = [duration_preds, durations_int, energy_preds, pitch_preds, input_lengths]
        decoder_preds = [mel_spec_preds, output_lengths]
        return encoder_preds, decoder_preds, weights
