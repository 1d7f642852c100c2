"""Hyper-parameters that shape the acoustic-model hot path.

Only the keys the model / loss / duration math read are kept (reference:
src/daft_exprt/hparams.py:36-127 for the defaults, model.py:838-856 and
loss.py:22-50 for the keys that are actually consumed).  File-system, MFA,
feature-extraction and trainer keys of the reference's ``HyperParams`` are out
of scope and deliberately absent.

Any object exposing the same attribute names (for instance the reference's
own ``HyperParams`` instance, or an ``argparse.Namespace`` built from a
``config.json``) can be handed to ``DaftExprt`` / ``DaftExprtLoss`` instead.
"""
from __future__ import annotations

import copy

N_SYMBOLS_ENGLISH = 76  # len(symbols_english), reference symbols.py:16-36 (pad '_' at index 0)


def _fft_stack(nb_blocks=4, hidden=128, heads=2, conv_channels=1024, kernel=3, attn_dropout=0.1, conv_dropout=0.1):
    return {
        'nb_blocks': nb_blocks,
        'hidden_embed_dim': hidden,
        'attn_nb_heads': heads,
        'attn_dropout': attn_dropout,
        'conv_kernel': kernel,
        'conv_channels': conv_channels,
        'conv_dropout': conv_dropout,
    }


class HyperParams:
    """Attribute bag with the reference's default values for the hot path."""

    def __init__(self, **kwargs):
        # mel / duration arithmetic (hparams.py:39-46)
        self.centered = False
        self.sampling_rate = 22050
        self.n_mel_channels = 80
        self.filter_length = 1024
        self.hop_length = 256
        # vocabulary / speakers (hparams.py:186-200)
        self.n_symbols = N_SYMBOLS_ENGLISH
        self.n_speakers = 2
        self.external_emb_dim = 192  # model.py:855
        # loss weights (hparams.py:71-88, loss.py:24-40)
        self.post_mult_weight = 1e-3
        self.mel_spec_weight = 1.0
        self.adv_max_weight = 1e-2
        self.warmup_steps = 10000
        self.energy_consistency_weight = 0.05
        self.pitch_consistency_weight = 0.15
        self.pitch_predictor_path = ''
        # optimiser / schedule (hparams.py:92-100; read by optim.FusedAdam, optim.update_learning_rate and train_steps.Trainer)
        self.accumulation_steps = 1
        self.betas = [0.9, 0.98]
        self.epsilon = 1e-9
        self.weight_decay = 1e-6
        self.grad_clip_thresh = float('inf')
        self.initial_learning_rate = 1e-4
        self.max_learning_rate = 1e-3
        # gradient reversal strength (model.py:51; not defined by the reference defaults)
        self.lambda_reversal = 1.0
        # module shapes (hparams.py:106-127)
        self.phoneme_encoder = _fft_stack()
        self.gaussian_upsampling_module = {'conv_kernel': 3}
        fd = _fft_stack()
        del fd['hidden_embed_dim']  # inserted by the decoder itself, model.py:534
        self.frame_decoder = fd
        # per-speaker statistics used by pitch_shift (model.py:984-985)
        self.stats = {}
        for key, value in kwargs.items():
            setattr(self, key, value)

    def clone(self, **overrides):
        new = copy.deepcopy(self)
        for key, value in overrides.items():
            setattr(new, key, value)
        return new

    def without_dropout(self):
        new = copy.deepcopy(self)
        for name in ('phoneme_encoder', 'frame_decoder', 'accent_encoder'):
            cfg = getattr(new, name, None)
            if cfg is not None:
                cfg['attn_dropout'] = 0.0
                cfg['conv_dropout'] = 0.0
        return new
