"""Pins the CPU oracle against outputs of the reference implementation (tests/golden, written by make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import daft_exprt_oracle as oracle
from tests import helpers

TRAIN_CASES = ['train_halo', 'train_zero_dur', 'train_single', 'train_no_postmult']


def _hp_for(name):
    if name == 'train_no_postmult':
        return helpers.golden_hparams(post_mult_weight=0.0, energy_consistency_weight=0.0, pitch_consistency_weight=0.0)
    return helpers.golden_hparams()


def _sd_for(name, requires_grad=False):
    drop = ('style_adapter.post_multipliers',) if name == 'train_no_postmult' else ()
    sd = helpers.golden_state_dict(drop)
    if requires_grad:
        for v in sd.values():
            v.requires_grad_(True)
    return sd


@pytest.mark.parametrize('name', TRAIN_CASES)
def test_forward_and_loss(name):
    case = helpers.load_case(name)
    hp, sd = _hp_for(name), _sd_for(name)
    inputs, targets = helpers.case_inputs(case)
    with torch.no_grad():
        outputs, internals = oracle.forward(sd, inputs, hp, return_internals=True)
        total, terms = oracle.loss(outputs, targets, int(case['meta/iteration']), hp, helpers.golden_pitch_predictor_state_dict())
    spk_preds, film, _, (mel, _), weights = outputs
    assert mel.shape == case['out/mel'].shape and weights.shape == case['out/weights'].shape
    np.testing.assert_allclose(mel.numpy(), case['out/mel'], rtol=0, atol=2e-5)
    assert np.abs(mel.numpy() - case['out/mel']).mean() < 2e-6
    np.testing.assert_allclose(weights.numpy(), case['out/weights'], rtol=0, atol=5e-5)
    np.testing.assert_allclose(spk_preds.numpy(), case['out/speaker_preds'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(film[3].numpy(), case['out/film_dec'], rtol=0, atol=1e-5)
    for k, v in internals.items():
        np.testing.assert_allclose(v.numpy(), case['int/' + k], rtol=5e-5, atol=2e-5, err_msg=k)
    assert abs(total.item() - float(case['loss/total'])) <= 1e-5 * abs(float(case['loss/total']))
    for k, v in terms.items():
        ref = float(case['loss/' + k])
        assert abs(float(v) - ref) <= 1e-5 * max(1.0, abs(ref)), k
    # padded outputs are exactly zero
    out_l = inputs[9]
    for b in range(mel.shape[0]):
        assert (mel[b, :, int(out_l[b]):] == 0).all()


@pytest.mark.parametrize('name', ['train_halo', 'train_single'])
def test_gradients(name):
    case = helpers.load_case(name)
    hp, sd = _hp_for(name), _sd_for(name, requires_grad=True)
    inputs, targets = helpers.case_inputs(case)
    outputs = oracle.forward(sd, inputs, hp, training=True)
    total, _ = oracle.loss(outputs, targets, int(case['meta/iteration']), hp, helpers.golden_pitch_predictor_state_dict())
    total.backward()
    for k, p in sd.items():
        assert p.grad is not None, k
        s, a, smp = helpers.sample_like_golden(p.grad)
        ref_a = float(case['grad_abs/' + k])
        assert abs(a - ref_a) <= 2e-4 * max(ref_a, 1e-6), (k, a, ref_a)
        ref_smp = case['grad_smp/' + k]
        scale = max(float(np.abs(ref_smp).max()), 1e-8)
        assert np.abs(smp - ref_smp).max() <= 2e-4 * scale + 1e-7, k


@pytest.mark.parametrize('name,transform', [('inference_add', 'add'), ('inference_multiply', 'multiply')])
def test_inference(name, transform):
    case = helpers.load_case(name)
    hp = helpers.golden_hparams(stats={'spk 0': {'pitch': {'mean': 5.0, 'std': 0.25}},
                                       'spk 1': {'pitch': {'mean': 4.6, 'std': 0.3}}})
    sd = helpers.golden_state_dict()
    t = lambda k: torch.from_numpy(case[k]).clone()
    inputs = (t('in/symbols'), t('in/dur_factors'), t('in/energy_factors'), t('in/pitch_factors'),
              t('in/input_lengths'), t('in/speaker_ids'))
    prosody = {k: t('in/prosody_' + k) for k in ('duration_preds', 'durations_int', 'energy_preds', 'pitch_preds')}
    with torch.no_grad():
        enc, (mel, out_lens), weights = oracle.inference(sd, inputs, transform, hp, external_prosody=prosody,
                                                         external_embeddings=t('in/spk_embs'),
                                                         external_accent_emb=t('in/accent_emb'))
    assert np.array_equal(enc[1].numpy(), case['out/durations_int'])          # bit-exact integer path
    assert np.array_equal(out_lens.numpy(), case['out/output_lengths'])
    assert mel.shape == case['out/mel'].shape
    np.testing.assert_allclose(enc[0].numpy(), case['out/duration_preds'], rtol=0, atol=0)
    np.testing.assert_allclose(enc[2].numpy(), case['out/energy_preds'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(enc[3].numpy(), case['out/pitch_preds'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(mel.numpy(), case['out/mel'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(weights.numpy(), case['out/weights'], rtol=0, atol=5e-5)


def test_duration_known_answers():
    with open(os.path.join(helpers.GOLDEN, 'duration_kats.json')) as f:
        kats = json.load(f)
    hp = helpers.golden_hparams()
    for kat in kats['duration_to_integer']:
        hpk = hp.clone(centered=True) if kat.get('centered') else hp
        try:
            got = oracle.duration_to_integer([list(s) for s in kat['spans']], hpk)
        except (IndexError, ValueError) as exc:
            got = type(exc).__name__
        assert got == kat['expected'], kat
    g = kats['get_int_durations']
    f_out, i_out = oracle.get_int_durations(torch.tensor(g['input'], dtype=torch.float32), hp)
    assert i_out.tolist() == g['int_out']
    assert f_out.tolist() == g['float_out']


def test_error_behaviour():
    hp, sd = helpers.golden_hparams(), helpers.golden_state_dict()
    inputs, _ = helpers.case_inputs(helpers.load_case('train_single'))
    with pytest.raises(ValueError):
        oracle.forward(sd, inputs[:11], hp)
    with pytest.raises(ValueError):
        oracle.forward(sd, inputs[:11] + (None,), hp)


def test_batch_conditioning_vs_reference():
    """oracle.process_batch against DynamicSpeakerStatsManager.process_batch outputs (f-2)."""
    case = helpers.load_case('batch_conditioning')
    inputs = tuple(torch.from_numpy(case['in/' + n]) for n in helpers.INPUT_NAMES)
    stats = {}
    for k in case:
        if k.startswith('stats/'):
            s = int(k.split('/')[1])
            v = case[k]
            stats[s] = {'energy': {'mean': float(v[0]), 'std': float(v[1])}, 'pitch': {'mean': float(v[2]), 'std': float(v[3])},
                        'spk_emb': torch.from_numpy(case[f'emb/{s}'])}
    out = oracle.process_batch(inputs, stats)
    for n, t in zip(helpers.INPUT_NAMES, out):
        assert np.array_equal(t.numpy(), case['out/' + n]), n
