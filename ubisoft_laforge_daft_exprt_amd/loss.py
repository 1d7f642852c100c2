"""``DaftExprtLoss`` with the reference's interface (src/daft_exprt/loss.py:11-159), computed by fused HIP reductions.

``DaftExprtLoss(device, hparams)(outputs, targets, iteration) -> (loss, dict of 7 floats)``.  All seven terms are reduced
on the device and fetched with ONE host transfer (the reference issues seven ``.item()`` syncs, loss.py:149-157).
The gradient w.r.t. the predicted mel, the speaker logits and the post-multipliers is produced in the same pass and
handed to autograd by ``_LossFn``.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import ops
from .functional import Lengths


_PITCH_16BIT = os.environ.get('DX_PITCH_16BIT', '1') != '0'
_PITCH_FUSED = os.environ.get('DX_PITCH_FUSED', '1') != '0'      # 0: the layer-by-layer launches also in the 16-bit modes (A/B timing, tests)


def pitch_predictor_shapes(n_mel_channels=80, hidden_dim=256, kernel_size=3):
    """state-dict keys / shapes of the frozen PitchPredictor checkpoint (layers/pitch_predictor.py:38-74: three weight-normed
    Conv1d + BatchNorm1d stages and a weight-normed Conv1d to one channel), for seeded synthetic weights."""
    shapes = {}
    dims = [(n_mel_channels, hidden_dim), (hidden_dim, hidden_dim), (hidden_dim, hidden_dim), (hidden_dim, 1)]
    for i, (cin, cout) in enumerate(dims):
        pre = f'conv_layers.{4 * i}.conv.'
        shapes[pre + 'bias'] = (cout,)
        shapes[pre + 'weight_g'] = (cout, 1, 1)
        shapes[pre + 'weight_v'] = (cout, cin, kernel_size)
        if i < 3:
            bn = f'conv_layers.{4 * i + 2}.'
            for name in ('weight', 'bias', 'running_mean', 'running_var'):
                shapes[bn + name] = (cout,)
            shapes[bn + 'num_batches_tracked'] = ()
    return shapes


def fold_pitch_predictor(state_dict, device, rt=None):
    """Frozen PitchPredictor (layers/pitch_predictor.py:38-74) -> plain conv weights: weight_norm (w = g v / ||v||, norm over
    (in, k) per output channel) folded into the weight, eval-mode BatchNorm1d folded into a per-channel scale/shift."""
    layers = []
    for conv_idx, bn_idx in ((0, 2), (4, 6), (8, 10), (12, None)):
        pre = f'conv_layers.{conv_idx}.conv.'
        if pre + 'weight_v' in state_dict:
            v, g = state_dict[pre + 'weight_v'].float(), state_dict[pre + 'weight_g'].float()
            w = v * (g / v.norm(dim=(1, 2), keepdim=True))
        elif pre + 'parametrizations.weight.original1' in state_dict:
            v, g = state_dict[pre + 'parametrizations.weight.original1'].float(), state_dict[pre + 'parametrizations.weight.original0'].float()
            w = v * (g / v.norm(dim=(1, 2), keepdim=True))
        else:
            w = state_dict[pre + 'weight'].float()
        b = state_dict[pre + 'bias'].float()
        layer = {'w': w.to(device).contiguous(), 'b': b.to(device).contiguous(), 'scale': None, 'shift': None}
        if bn_idx is not None:
            bn = f'conv_layers.{bn_idx}.'
            scale = state_dict[bn + 'weight'].float() / torch.sqrt(state_dict[bn + 'running_var'].float() + 1e-5)
            shift = state_dict[bn + 'bias'].float() - state_dict[bn + 'running_mean'].float() * scale
            layer['scale'], layer['shift'] = scale.to(device).contiguous(), shift.to(device).contiguous()
            layer['zeros'] = torch.zeros_like(layer['shift'])
        layers.append(layer)
    # the last conv has ONE output channel; the kernels want Cout % 4 == 0 -> zero-pad to 4 rows
    last = layers[-1]
    wpad = torch.zeros(4, *last['w'].shape[1:], device=device)
    wpad[:1] = last['w']
    bpad = torch.zeros(4, device=device)
    bpad[:1] = last['b']
    last['b3'] = float(last['b'][0])        # (host copy of the one bias the fused chain takes by value; read once here, at load time)
    last['w'], last['b'] = wpad, bpad
    for layer in layers:
        layer['pack'] = ops.PackedWeight(layer['w'], rt)
    return layers


class _LossFn(torch.autograd.Function):
    """total = w_spk * CE + pmw * ||pm||_2 + msw * (L1 + L2) + ecw * E + pcw * P; gradients computed in the forward."""

    @staticmethod
    def forward(ctx, mel_pred, speaker_preds, post_multipliers, mel_target, speaker_ids, frames_pitch, lens: Lengths, cfg, pitch_layers, rt=None):
        dev = mel_pred.device
        B, M, T = mel_pred.shape
        mel_pred, mel_target = mel_pred.contiguous(), mel_target.contiguous()
        arena = None if rt is None else rt.arena
        ce = dlogits = None
        if speaker_preds is not None:
            ce, dlogits = ops.cross_entropy(speaker_preds.contiguous(), speaker_ids.contiguous())
        pm = post_multipliers.detach().contiguous() if post_multipliers is not None else None      # 16 numbers
        ep, et, sums = ops.mel_stats(mel_pred, mel_target, arena=arena)
        des = esum = None
        if cfg['ecw'] > 0:
            des, esum = ops.energy_diff(ep, et, lens.i32, arena=arena)
        # energy term: ecw / (sum of valid lengths), the division done on the device (nothing host-side is frozen into a graph)
        need_grad = cfg.get('need_grad', True)       # False under no_grad (Trainer.validate): the terms only, none of the gradient launches
        # ``grad_scale``: d(what the caller differentiates) / d(total), known up front (a trainer: loss scale / accumulation steps).  Every
        # gradient produced here is multiplied by it on the way out and ``backward`` then hands them on as they are: no ``total * c`` launch,
        # no three ``* g_total`` launches (one of them a 14 MB read-modify-write of the mel gradient).
        gs = float(cfg.get('grad_scale', 1.0))
        dmel = ops.mel_grad(mel_pred, mel_target, ep, des, lens.i32, gs * cfg['msw'] / (M * B), gs * cfg['msw'] / (M * B),
                            gs * cfg['ecw'] if des is not None else 0.0, e_per_total=True) if need_grad else None
        psum = None
        if pitch_layers is not None and frames_pitch is not None and cfg['pcw'] > 0:
            prec = pitch_layers[0]['pack'].rt.precision                          # one value for the whole chain
            frames_pitch = frames_pitch.contiguous()
            if _PITCH_FUSED and ops.pitch_chain_applies(pitch_layers, mel_pred, prec):
                # one launch each way (csrc/dx_pitch.hip): activations stay in LDS, the backward gets ReLU sign bits instead of activations
                pp, masks = ops.pitch_chain_fwd(mel_pred, pitch_layers, lens.i32, prec, arena=arena)
                psum = ops.pitch_mse(pp, frames_pitch, lens.i32, arena=arena)
                if need_grad:
                    g = ops.pitch_grad(pp, frames_pitch, lens.i32, psum, gs * cfg['pcw'], out=ops._zeros(arena, B, T, device=dev))
                    ops.pitch_chain_bwd(g, masks, pitch_layers, lens.i32, prec, dmel)              # dmel is this function's own fresh tensor
            else:
                # layer by layer (the exact-fp32 mode, other architectures): channels-last; gradient flows through it to the mel only
                x = ops.transpose(mel_pred)                                          # (B, T, M)
                acts = []
                depth = len(pitch_layers) - 1                                       # stacked k=3 convs: halos 3, 2, 1, 0
                hd = ops.hidden_dtype(prec) if _PITCH_16BIT else torch.float32     # 16-bit modes: the 256-wide activations of the frozen predictor
                for i, layer in enumerate(pitch_layers[:-1]):                      # (only ever GEMM operands / ReLU masks) are stored in 16 bits
                    r = ops.conv_gemm(x, layer['pack'], layer['b'], relu=True, lens=lens.i32, halo=depth - i, prec=prec, out_dtype=hd)
                    acts.append(r)
                    x = ops.channel_affine(r, layer['scale'], layer['shift'], prec=prec)
                last = pitch_layers[-1]
                pp = ops.conv_gemm(x, last['pack'], last['b'], lens=lens.i32, halo=0, prec=prec)      # (B, T, 4): channel 0 is the prediction
                psum = ops.pitch_mse(pp, frames_pitch, lens.i32, arena=arena)                          # (read and written in place: no slice copies)
                if need_grad:
                    g = ops.pitch_grad(pp, frames_pitch, lens.i32, psum, gs * cfg['pcw'], out=ops._zeros(arena, B, T, 4, device=dev))
                    # each input-gradient GEMM applies the previous layer's BatchNorm scale and ReLU mask in its epilogue
                    for k in range(len(pitch_layers) - 1, 0, -1):
                        prev = pitch_layers[k - 1]
                        g = ops.conv_gemm(g, pitch_layers[k]['pack'], None, transpose=True, post_scale=prev['scale'], post_shift=prev['zeros'],
                                          relu_aux=acts[k - 1], lens=lens.i32, halo=depth - k + 1, prec=prec, out_dtype=hd)
                    d = ops.conv_gemm(g, pitch_layers[0]['pack'], None, transpose=True, lens=lens.i32, halo=0, prec=prec)
                    dmel = ops.transpose(d, add_to=dmel)                                   # dmel is this function's own fresh tensor
        # the seven terms, the total and the two small gradients: one launch (was ~30 one-element ATen launches)
        terms, total, d_spk, d_pm = ops.loss_finalize(ce, dlogits, cfg['spk_weight'], pm, cfg['pmw'], sums, lens.i32, M, cfg['msw'],
                                                      esum, cfg['ecw'], psum, cfg['pcw'], grad_scale=gs)
        if d_pm is not None:
            d_pm = d_pm.view_as(post_multipliers)
        if need_grad:
            ctx.save_for_backward(dmel, d_spk, d_pm)
        ctx.prescaled = 'grad_scale' in cfg
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)
        return total, terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        dmel, d_spk, d_pm = ctx.saved_tensors
        if ctx.prescaled:                          # the caller's factor is already inside (and g_total is its constant 1)
            return (dmel, d_spk, d_pm, None, None, None, None, None, None, None)
        return (dmel * g_total, None if d_spk is None else d_spk * g_total, None if d_pm is None else d_pm * g_total,
                None, None, None, None, None, None, None)


class DaftExprtLoss(nn.Module):
    def __init__(self, device, hparams):
        super().__init__()
        self.device = device
        self.nb_channels = hparams.n_mel_channels
        self.warmup_steps = getattr(hparams, 'warmup_steps', 10000)
        self.adv_max_weight = getattr(hparams, 'adv_max_weight', 1e-2)
        self.post_mult_weight = getattr(hparams, 'post_mult_weight', 1e-3)
        self.mel_spec_weight = getattr(hparams, 'mel_spec_weight', 1.0)
        self.energy_consistency_weight = getattr(hparams, 'energy_consistency_weight', 0.0)
        self.pitch_consistency_weight = getattr(hparams, 'pitch_consistency_weight', 0.0)
        self.pitch_layers = None
        # None: ``backward`` multiplies by the incoming gradient, as autograd expects.  A float (set by trainer.Trainer): the loss's gradients
        # are produced pre-multiplied by it and the trainer calls ``loss.backward(gradient=1)``
        self.grad_scale = None
        self.runtime = ops.Runtime(ops.DEFAULT.precision)     # this object's own execution state (see ops.Runtime)
        pp_path = getattr(hparams, 'pitch_predictor_path', '')
        if self.pitch_consistency_weight > 0 and pp_path:
            state = torch.load(pp_path, map_location='cpu', weights_only=True)
            self.load_pitch_predictor(state)

    def load_pitch_predictor(self, state_dict):
        """Frozen predictor from an in-memory state dict (same keys as layers/pitch_predictor.py)."""
        self.pitch_layers = fold_pitch_predictor(state_dict, self.device, self.runtime)

    def set_precision(self, name: str):
        self.runtime.set_precision(name)
        return self

    def update_adversarial_weight(self, iteration):
        """loss.py:52-55"""
        weight_iter = iteration * self.warmup_steps ** -1.5 * self.adv_max_weight / self.warmup_steps ** -0.5
        return min(self.adv_max_weight, weight_iter)

    def forward(self, outputs, targets, iteration):
        if len(targets) == 8:
            _, _, _, mel_targets, output_lengths, speaker_ids, _frames_energy, frames_pitch = targets
        else:
            _, _, _, mel_targets, output_lengths, speaker_ids = targets
            frames_pitch = None
        speaker_preds, film_params, _, decoder_preds, _ = outputs
        post_multipliers = film_params[0]
        mel_preds, output_lengths = decoder_preds
        if not mel_preds.is_cuda:
            raise RuntimeError('DaftExprtLoss (MI355X build) runs on the GPU only; there is no CPU path')
        lens = output_lengths if isinstance(output_lengths, Lengths) else getattr(output_lengths, '_dx_lengths', None)
        if lens is None or lens.i64 is not output_lengths:     # (the model leaves its Lengths object on the tensor it returns)
            lens = Lengths(output_lengths, host=getattr(output_lengths, '_dx_host_lengths', None))
        pm = post_multipliers if (self.post_mult_weight != 0.0 and torch.is_tensor(post_multipliers)) else None
        # ``iteration``: the step number, or -- from a trainer that replays captured graphs -- the adversarial weight itself as a
        # device scalar it updates before every replay
        spk_weight = iteration if torch.is_tensor(iteration) else self.update_adversarial_weight(iteration)
        cfg = {'need_grad': torch.is_grad_enabled() and mel_preds.requires_grad,
               **({'grad_scale': self.grad_scale} if self.grad_scale is not None else {}),
               'spk_weight': spk_weight, 'pmw': self.post_mult_weight, 'msw': self.mel_spec_weight,
               'ecw': self.energy_consistency_weight, 'pcw': self.pitch_consistency_weight if self.pitch_layers is not None else 0.0}
        total, terms = _LossFn.apply(mel_preds, speaker_preds, pm, mel_targets, speaker_ids, frames_pitch, lens, cfg, self.pitch_layers, self.runtime)
        return total, LossTerms(terms)


class LossTerms(dict):
    """The reference's ``individual_loss`` dict of 7 floats (loss.py:149-157).  The values live in one device tensor and are
    fetched with a single host transfer on FIRST ACCESS, so a training loop that only logs every n-th step never stalls
    the stream between forward and backward (the reference does seven ``.item()`` syncs per step)."""

    KEYS = ('speaker_loss', 'speaker_ce_raw', 'post_mult_loss', 'mel_spec_l1_loss', 'mel_spec_l2_loss',
            'energy_consistency_loss', 'pitch_consistency_loss')

    def __init__(self, device_terms):
        super().__init__()
        self._device_terms = device_terms

    def _fetch(self):
        if self._device_terms is not None:
            values = self._device_terms.tolist()
            self._device_terms = None
            super().update(zip(self.KEYS, values))

    def __getitem__(self, k):
        self._fetch()
        return super().__getitem__(k)

    def __iter__(self):
        self._fetch()
        return super().__iter__()

    def __len__(self):
        return len(self.KEYS)

    def __contains__(self, k):
        return k in self.KEYS

    def keys(self):
        self._fetch()
        return super().keys()

    def items(self):
        self._fetch()
        return super().items()

    def values(self):
        self._fetch()
        return super().values()

    def get(self, k, default=None):
        self._fetch()
        return super().get(k, default)

    def __repr__(self):
        self._fetch()
        return super().__repr__()
