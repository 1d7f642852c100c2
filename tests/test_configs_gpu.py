"""Parity of the HIP path ON THE BASELINE.json WORKLOAD SHAPES (C2 training batch, C4 inference batch) and on
tile-boundary lengths, against the CPU oracle, in BOTH operand modes.

  * C2, exact-f32 operands: the BASELINE.json bar (valid-frame mel L1 <= 1e-4; asserted 2e-5), 7 loss terms, all 184 gradients
  * C2, bf16 operands (the mode bench.py measures): STATED tolerances below -- the reference has no reduced precision
    (SURVEY.md §5), so these bounds are the build's own; measured values are recorded in DESIGN.md §2
  * C4: B = 256 inference; integer paths exact for every utterance, mel against the oracle on a 64-utterance sub-batch that keeps
    the full batch's padded geometry (the longest utterance on each axis rides along, so L_max / T_max and every halo are equal)
  * tile edges: lengths 126..130 and 254..257 on both axes, so that every "is the next 128-token tile live?" decision of the
    conv / LayerNorm / attention kernels is exercised at len % 128 in {126, 127, 0, 1, 2}

Dropout p = 0 everywhere (torch's RNG stream cannot be matched: SURVEY.md §7); train mode, so the backward is the real one.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ITERATION = 4000

# ---- stated 16-bit-operand tolerances (throughput modes; fp32 accumulate, 16-bit MFMA operands, 16-bit-stored 1024-wide tensors) ----
# Each bar is <= 1.5 x the value measured on MI355X at C2 (profiles/r03_parity_c2_*.json).  None of them is the 1e-4
# north-star bar: that one is met by the f32 mode only (test_c2_f32_* below); bench.py prints this next to the throughput.
BARS = {
    #            valid-frame mean |mel - oracle|; each loss term, relative; per-parameter max-norm relative gradient error;
    #            cosine with the oracle gradient (>= 64 elements); cosine for tensors below 64 elements
    'bf16': {'mel_l1': 8.4e-3, 'loss_rel': 6e-3, 'grad_rel': 0.085, 'grad_cos': 0.9972, 'grad_cos_small': 0.97},    # measured 5.6e-3, 3.8e-3, 5.0e-2 .. 5.6e-2, 0.9981 .. 0.9984
    'fp16': {'mel_l1': 1.2e-3, 'loss_rel': 9e-4, 'grad_rel': 0.050, 'grad_cos': 0.9995, 'grad_cos_small': 0.97},    # measured 7.9e-4, 5.7e-4, 2.7e-2 .. 3.3e-2, 0.9997
}


def valid_mel_l1(mel, ref, out_lens):
    tot, cnt = 0.0, 0
    for b, n in enumerate(out_lens.tolist()):
        tot += np.abs(mel[b, :, :n] - ref[b, :, :n]).sum()
        cnt += mel.shape[1] * n
    return tot / cnt


def _dump(name, record):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), 'w') as f:
            json.dump(record, f, indent=1, default=float)
    except OSError:
        pass


def _oracle_step(batch, hp, n_threads=None):
    """One oracle forward + loss + backward on CPU -> (mel, weights, total, terms, grads)."""
    from oracle import daft_exprt_oracle as oracle
    if n_threads:
        torch.set_num_threads(n_threads)
    sd = helpers.golden_state_dict()
    for v in sd.values():
        v.requires_grad_(True)
    cpu_inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    cpu_targets = (batch[1], batch[3], batch[4], batch[8], batch[9], batch[10], batch[6], batch[7])
    out = oracle.forward(sd, cpu_inputs, hp, training=True)
    total, terms = oracle.loss(out, cpu_targets, ITERATION, hp, helpers.golden_pitch_predictor_state_dict())
    total.backward()
    return dict(mel=out[3][0].detach().numpy(), weights=out[4].detach().numpy(), total=float(total.detach()),
                terms={k: float(v) for k, v in terms.items()}, grads={k: v.grad.clone() for k, v in sd.items()},
                spk_preds=out[0].detach().numpy())


FP16_LOSS_SCALE = 4096.0      # the trainer's static loss scale of the fp16 mode (gradients that live in fp16 tensors would underflow)


def _hip_step(pkg, batch, hp, precision):
    pkg.set_precision(precision)
    try:
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        model.train()
        crit = pkg.DaftExprtLoss(DEV, hp)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
        inputs, targets = model.parse_batch(DEV, batch)
        out = model(inputs)
        total, terms = crit(out, targets + (inputs[6], inputs[7]), ITERATION)
        scale = FP16_LOSS_SCALE if precision == 'fp16' else 1.0
        (total * scale).backward()
        torch.cuda.synchronize()
        return dict(mel=out[3][0].detach().cpu().numpy(), weights=out[4].detach().cpu().numpy(), total=float(total),
                    terms={k: float(v) for k, v in terms.items()}, grads={k: p.grad.detach().cpu() / scale for k, p in model.named_parameters()},
                    spk_preds=out[0].detach().cpu().numpy())
    finally:
        pkg.set_precision('f32')


def _grad_metrics(got, ref):
    rows = {}
    for k, r in ref.items():
        g = got[k].double().flatten()
        r = r.double().flatten()
        rel = float((g - r).abs().max() / r.abs().max().clamp_min(1e-30))
        cos = float((g @ r) / (g.norm() * r.norm()).clamp_min(1e-30))
        rows[k] = (rel, cos, r.numel())
    return rows


@pytest.fixture(scope='module')
def pkg():
    import ubisoft_laforge_daft_exprt_amd as p
    p.set_precision('f32')
    return p


@pytest.fixture(scope='module')
def c2():
    """The C2 batch (BASELINE.json configs[1]: 48 utterances, L 50..120, T_max ~ 900) and ONE oracle step on it."""
    from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch
    hp = helpers.golden_hparams()
    batch = synthetic_batch(n_speakers=hp.n_speakers, **CONFIGS['C2'])
    return hp, batch, _oracle_step(batch, hp, n_threads=min(16, len(os.sched_getaffinity(0))))


def test_c2_f32_forward_loss_gradients_vs_oracle(pkg, c2):
    hp, batch, ref = c2
    got = _hip_step(pkg, batch, hp, 'f32')
    assert got['mel'].shape == ref['mel'].shape and got['weights'].shape == ref['weights'].shape
    l1 = valid_mel_l1(got['mel'], ref['mel'], batch[9])
    assert l1 < 2e-5, l1                                                    # BASELINE.json bar: 1e-4
    assert np.abs(got['weights'] - ref['weights']).max() < 2e-4
    assert np.abs(got['spk_preds'] - ref['spk_preds']).max() < 1e-4
    for b, n in enumerate(batch[9].tolist()):
        assert (got['mel'][b, :, n:] == 0).all()
    assert abs(got['total'] - ref['total']) <= 2e-5 * abs(ref['total'])
    for k, v in got['terms'].items():
        assert abs(v - ref['terms'][k]) <= 5e-5 * max(1.0, abs(ref['terms'][k])), (k, v, ref['terms'][k])
    rows = _grad_metrics(got['grads'], ref['grads'])
    worst = max(rows.items(), key=lambda kv: kv[1][0])
    _dump('parity_c2_f32.json', {'mel_l1': l1, 'worst_grad': [worst[0], worst[1][0]], 'loss_total': [got['total'], ref['total']]})
    print(f'C2 f32: valid mel L1 {l1:.3e}; worst gradient rel err {worst[1][0]:.3e} ({worst[0]})')
    for k, (rel, cos, _) in rows.items():
        assert rel < 3e-3 and cos > 0.99999, (k, rel, cos)


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_c2_bf16_forward_loss_gradients_vs_oracle(pkg, c2, precision):
    """The mode and shape bench.py measures (bf16) and the fp16 twin of BASELINE.json config 5, end to end against the oracle, with
    the tolerances stated at the top of this file (fp16 carries 3 more mantissa bits than bf16 and lands well inside them)."""
    hp, batch, ref = c2
    got = _hip_step(pkg, batch, hp, precision)
    assert got['mel'].shape == ref['mel'].shape
    assert np.isfinite(got['mel']).all() and all(torch.isfinite(g).all() for g in got['grads'].values())
    l1 = valid_mel_l1(got['mel'], ref['mel'], batch[9])
    for b, n in enumerate(batch[9].tolist()):
        assert (got['mel'][b, :, n:] == 0).all()
    term_err = {k: abs(v - ref['terms'][k]) / max(abs(ref['terms'][k]), 1e-8) for k, v in got['terms'].items()}
    rows = _grad_metrics(got['grads'], ref['grads'])
    big = {k: v for k, v in rows.items() if v[2] >= 64}
    small = {k: v for k, v in rows.items() if v[2] < 64}
    worst_rel = max(rows.items(), key=lambda kv: kv[1][0])
    worst_cos = min(big.items(), key=lambda kv: kv[1][1])
    bars = BARS[precision]
    _dump(f'parity_c2_{precision}.json', {'mel_l1': l1, 'bar_mel_l1': bars['mel_l1'], 'bars': bars, 'loss_terms_rel': term_err, 'loss_total': [got['total'], ref['total']],
                                  'worst_grad_rel': [worst_rel[0], worst_rel[1][0]], 'worst_grad_cos': [worst_cos[0], worst_cos[1][1]],
                                  'grads': {k: [v[0], v[1]] for k, v in rows.items()}})
    print(f'C2 {precision}: valid mel L1 {l1:.3e}; loss total {got["total"]:.5f} vs {ref["total"]:.5f}; worst term rel {max(term_err.values()):.3e}; '
          f'worst grad rel {worst_rel[1][0]:.3e} ({worst_rel[0]}); worst grad cos {worst_cos[1][1]:.5f} ({worst_cos[0]})')
    assert l1 < bars['mel_l1'], l1
    assert abs(got['total'] - ref['total']) <= bars['loss_rel'] * abs(ref['total'])
    for k, e in term_err.items():
        assert e <= bars['loss_rel'], (k, e)
    for k, (rel, cos, _) in big.items():
        assert cos >= bars['grad_cos'] and rel <= bars['grad_rel'], (k, rel, cos)
    for k, (rel, cos, _) in small.items():
        assert cos >= bars['grad_cos_small'], (k, rel, cos)


@pytest.mark.parametrize('precision,cuts', [('bf16', 3), ('fp16', 3), ('bf16', 0)])
def test_c2_trainer_gradient_path_equals_plain_autograd(pkg, c2, precision, cuts):
    """The gradients the TRAINER computes on the C2 batch - kernels accumulating straight into the all-reduce buckets (sink), the FFT
    blocks' weight gradients queued and launched eight layers at a time, the step arena, loss scaling in fp16 - against plain autograd
    (.grad tensors, one launch per weight gradient) of the same model, same mode, dropout off: same kernels, other summation orders.
    ``cuts`` = 3: the four-phase backward of an N > 1 step (three cuts, four exchange groups); 0: the single phase of one rank."""
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp, batch, _ = c2
    hp = hp.without_dropout() if hasattr(hp, 'without_dropout') else hp
    plain = _hip_step(pkg, batch, hp, precision)['grads']
    pkg.set_precision(precision)
    try:
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        crit = pkg.DaftExprtLoss(DEV, hp)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    finally:
        pkg.set_precision('f32')
    t = Trainer(model, crit, hp, use_graphs=False, cuts=cuts)
    assert model.runtime.sink and model.runtime.defer_wgrad
    assert sorted(set(t.reducer.bucket_group)) == list(range(cuts + 1))
    plan = t.exchange_plan()
    assert plan['total_bytes'] == 4 * sum(p.numel() for p in model.parameters())
    if cuts == 3:
        assert plan['exposed_fraction'] <= 0.25, plan       # what is launched after the last backward kernel
    t.iteration = ITERATION
    dev_batch = tuple(x.to(DEV) if torch.is_tensor(x) else x for x in batch)
    parsed, _ = t._parse([dev_batch])
    launched = []

    def launch(gid):
        launched.append(gid)
        t.reducer.launch_group(gid)
    _, _, rest = t._phases(parsed, ITERATION, launch)      # the eager form of train_step, without the optimiser
    assert len(rest) == cuts
    for run in rest:
        run()
    assert launched == list(range(cuts + 1))
    t.reducer.finish()
    torch.cuda.synchronize()
    assert not model.runtime.wgrad_queue
    scale = t.loss_scale
    worst = ('', 0.0)
    for k, prm in model.named_parameters():
        g = prm.grad.detach().cpu().double() / scale
        r = plain[k].double()
        rel = float((g - r).abs().max() / r.abs().max().clamp_min(1e-30))
        if rel > worst[1]:
            worst = (k, rel)
    print(f'C2 {precision}: trainer gradient path vs plain autograd, worst relative difference {worst[1]:.2e} at {worst[0]}')
    assert worst[1] < 2e-3, worst
    t.reducer.remove()


def test_c2_later_phases_with_bucket_traffic_on_another_queue(pkg, c2):
    """What an N > 1 step does on the device, on one GPU: while phases B, C, D (accent-encoder backward, its kernels adding into the
    later groups' buckets with memory-side float atomics) run on the compute stream, another queue reads and rewrites the buckets of the
    groups already finished in place, as their all-reduces do.  Every gradient must equal the serial step's (summation order only)."""
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp, batch, _ = c2
    hp = hp.without_dropout() if hasattr(hp, 'without_dropout') else hp
    pkg.set_precision('bf16')
    try:
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        crit = pkg.DaftExprtLoss(DEV, hp)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    finally:
        pkg.set_precision('f32')
    t = Trainer(model, crit, hp, use_graphs=False, cuts=3)
    t.iteration = ITERATION
    red = t.reducer
    assert sorted(set(red.bucket_group)) == [0, 1, 2, 3] and all(o % 1024 == 0 for o in red.bucket_offset)
    dev_batch = tuple(x.to(DEV) if torch.is_tensor(x) else x for x in batch)
    parsed, _ = t._parse([dev_batch])
    side = torch.cuda.Stream()

    def step(traffic):
        def launch(gid):
            red.launch_group(gid)
            if traffic and gid < 3:                               # "all-reduce" of the finished group beside the next phase
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(8):                            # several passes of in-place traffic, the length of the next phase
                        for bi, flat in enumerate(red.flat):
                            if red.bucket_group[bi] == gid:
                                flat.mul_(1.0)
        _, _, rest = t._phases(parsed, ITERATION, launch)
        for run in rest:
            run()
        torch.cuda.current_stream().wait_stream(side)
        red.finish()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    serial = step(False)
    for _ in range(3):
        got = step(True)
        worst = max(((float((got[k] - serial[k]).abs().max() / serial[k].abs().max().clamp_min(1e-30)), k) for k in serial))
        # fp32 atomics land in another order from run to run; a value that then becomes a 16-bit GEMM operand can round the other way
        # (seen: 3.8e-4 on one parameter, once).  A lost update of one workgroup's contribution would be 1e-2.
        assert worst[0] < 2e-3, worst
    print(f'C2 bf16: later phases beside bucket traffic on another queue vs serial, worst relative difference {worst[0]:.2e} at {worst[1]}')
    red.remove()


def test_c2_gradients_with_decoder_weight_gradients_on_a_side_stream(pkg, c2):
    """Round 2's "lost update", as a regression test.  With the frame decoder's queued k = 1 weight gradients (an MFMA kernel) launched on
    a SIDE stream beside the upsampler's backward, the energy / pitch projection gradients came out wrong on the middle tap of the even
    channels >= 64 in about half of the steps (1e-2 of the sum).  Round 3 reproduced it (tools/experiment_fork_wgrad.py) and bisected it
    to the packed form the compiler chose for that tap in ``scalar_conv_wgrad_kernel`` (v_pk_fma_f32 ... op_sel:[0,1,0]); written as single
    v_fma_f32 instructions the kernel is exact beside any other kernel (24 of 24 steps, against 11 of 24).  Here: 8 forked steps, every
    gradient equal to the serial step's."""
    from ubisoft_laforge_daft_exprt_amd import ops
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp, batch, _ = c2
    hp = hp.without_dropout() if hasattr(hp, 'without_dropout') else hp
    pkg.set_precision('bf16')
    try:
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        crit = pkg.DaftExprtLoss(DEV, hp)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    finally:
        pkg.set_precision('f32')
    t = Trainer(model, crit, hp, use_graphs=False, cuts=0)
    rt = model.runtime
    dev_batch = tuple(x.to(DEV) if torch.is_tensor(x) else x for x in batch)
    parsed, _ = t._parse([dev_batch])
    side = torch.cuda.Stream()
    state = {'fork': False, 'keep': []}
    original = ops.upsample_bwd

    def upsample_bwd(*a, **kw):               # first call of the upsampler's backward: the decoder's backward has been issued
        if state['fork']:
            k1 = {k: v for k, v in rt.wgrad_queue.items() if k[0] == 1}
            for k in k1:
                del rt.wgrad_queue[k]
            state['keep'].append(k1)          # the operands stay referenced until the streams are joined
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rest, rt.wgrad_queue = rt.wgrad_queue, dict(k1)
                ops.flush_wgrads(rt)
                rt.wgrad_queue = rest
        return original(*a, **kw)

    def step(fork):
        state['fork'] = fork
        t._phases(parsed, ITERATION, t.reducer.launch_group)
        torch.cuda.current_stream().wait_stream(side)
        state['keep'].clear()
        t.reducer.finish()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    ops.upsample_bwd = upsample_bwd
    try:
        serial = step(False)
        for _ in range(8):
            got = step(True)
            worst = max(((float((got[k] - serial[k]).abs().max() / serial[k].abs().max().clamp_min(1e-30)), k) for k in serial))
            assert worst[0] < 1e-3, worst
    finally:
        ops.upsample_bwd = original
        t.reducer.remove()
    print(f'C2 bf16: decoder k = 1 weight gradients on a side stream vs serial, worst relative difference {worst[0]:.2e} at {worst[1]}')


# ----------------------------------------------------------------------------------------------------------------------
# C4: B = 256 inference
# ----------------------------------------------------------------------------------------------------------------------
def _c4(device):
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_inference_batch
    inputs, prosody, spk, accent = synthetic_inference_batch()
    mv = (lambda t: t.clone().to(device))
    return tuple(mv(t) for t in inputs), {k: mv(v) for k, v in prosody.items()}, mv(spk), mv(accent)


@pytest.mark.parametrize('precision,tol', [('f32', 2e-5), ('bf16', 6e-3), ('fp16', 1e-3)])      # measured 7.3e-7, 3.9e-3, 6.6e-4
def test_c4_inference_b256_vs_oracle(pkg, precision, tol):
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.inference import GraphedSynthesizer
    hp = helpers.golden_hparams(stats={'spk 0': {'pitch': {'mean': 5.0, 'std': 0.25}}, 'spk 1': {'pitch': {'mean': 4.6, 'std': 0.3}}})
    pkg.set_precision(precision)
    try:
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        model.eval()
        inputs, prosody, spk, accent = _c4(DEV)
        with torch.no_grad():
            enc, (mel, out_lens), weights = model.inference(inputs, 'add', hp, external_prosody=prosody, external_embeddings=spk,
                                                            external_accent_emb=accent)
        synth = GraphedSynthesizer(model, hp)                                  # the captured-graph form of the same call (f-3)
        inputs2, prosody2, spk2, accent2 = _c4(DEV)
        enc_g, (mel_g, len_g), w_g = synth(inputs2, 'add', prosody2, spk2, accent2, use_graph=True)
        assert torch.equal(mel, mel_g) and torch.equal(out_lens, len_g) and torch.equal(weights, w_g)
    finally:
        pkg.set_precision('f32')
    B = mel.shape[0]
    assert B == 256
    # integer paths: every utterance, bit-exact, against the oracle's double-precision host arithmetic
    cin, cpros, cspk, cacc = _c4('cpu')
    dur_ref, dur_int_ref = oracle.get_int_durations(cpros['duration_preds'] * cin[1], hp)
    assert torch.equal(enc[1].cpu(), dur_int_ref)
    totals = dur_int_ref.sum(dim=1)
    assert torch.equal(out_lens.cpu(), totals.clamp_min(1))
    assert mel.shape == (B, 80, int(totals.max())) and weights.shape == (B, cin[0].shape[1], int(totals.max()))
    # mel / alignment weights: a 64-utterance sub-batch with the SAME padded geometry (row 0 is the longest on the symbol axis,
    # `tmax_row` the longest on the frame axis), so the oracle's B*L*D*T broadcast stays at 2.6 GB
    tmax_row = int(totals.argmax())
    rows = sorted(set([0, tmax_row] + list(range(1, B, 4))))[:64]
    idx = torch.tensor(rows)
    sub_in = tuple(t[idx] for t in cin)
    sub_pros = {k: v[idx] for k, v in cpros.items()}
    sd = helpers.golden_state_dict()
    with torch.no_grad():
        enc_r, (mel_r, len_r), w_r = oracle.inference(sd, sub_in, 'add', hp, external_prosody=sub_pros, external_embeddings=cspk[idx],
                                                      external_accent_emb=cacc[idx])
    assert mel_r.shape[2] == mel.shape[2] and w_r.shape[1:] == weights.shape[1:]
    got = mel.cpu()[idx].numpy()
    l1 = valid_mel_l1(got, mel_r.numpy(), len_r)
    werr = np.abs(weights.cpu()[idx].numpy() - w_r.numpy()).max()
    print(f'C4 {precision}: B=256 T_max={mel.shape[2]} valid mel L1 vs oracle (64 utterances) {l1:.3e}; alignment max err {werr:.2e}')
    _dump(f'parity_c4_{precision}.json', {'mel_l1': l1, 'weights_max_err': float(werr), 'T_max': mel.shape[2]})
    assert l1 < tol, l1
    wmean = float(np.abs(weights.cpu()[idx].numpy() - w_r.numpy()).mean())
    # alignment weights are in [0, 1]; a narrow Gaussian (range ~ 1e-3 .. 0.3 frames) turns a bf16-sized change of its range into a
    # large change of a few individual weights, so the bf16 bound is on the mean and a loose one on the maximum
    assert werr < (2e-4 if precision == 'f32' else 0.35) and wmean < (1e-6 if precision == 'f32' else 2e-4), (werr, wmean)
    np.testing.assert_allclose(enc[3].cpu()[idx].numpy(), enc_r[3].numpy(), rtol=0, atol=5e-6)     # shifted pitch


# ----------------------------------------------------------------------------------------------------------------------
# tile-boundary lengths on both axes (f32)
# ----------------------------------------------------------------------------------------------------------------------
EDGE_LENS = [257, 256, 255, 254, 130, 129, 128, 127, 126]


def _edge_batch(axis, max_len):
    """axis 'frame': output lengths = EDGE_LENS (<= max_len) with ~40 symbols each; axis 'symbol': input lengths = EDGE_LENS."""
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    lens = [v for v in EDGE_LENS if v <= max_len]
    g = torch.Generator().manual_seed(77 + max_len)
    if axis == 'symbol':
        L = max(lens)
        dur = torch.randint(1, 3, (len(lens), L), generator=g)
        return synthetic_batch(len(lens), (1, L), seed=78, n_speakers=3, sym_lengths=lens, durations_int=dur)
    n_sym = list(range(44, 44 - len(lens), -1))                                # input lengths must be sorted descending
    L = max(n_sym)
    dur = torch.zeros(len(lens), L, dtype=torch.long)
    for b, (ns, total) in enumerate(zip(n_sym, lens)):
        base = torch.full((ns,), total // ns, dtype=torch.long)
        base[torch.randperm(ns, generator=g)[:total % ns]] += 1                # sums exactly to the target frame count
        dur[b, :ns] = base
    return synthetic_batch(len(lens), (1, L), seed=79, n_speakers=3, sym_lengths=n_sym, durations_int=dur)


@pytest.mark.parametrize('axis', ['frame', 'symbol'])
@pytest.mark.parametrize('max_len', [257, 256, 129, 128])
def test_tile_edge_lengths_f32_vs_oracle(pkg, axis, max_len):
    hp = helpers.golden_hparams()
    batch = _edge_batch(axis, max_len)
    lens = batch[9] if axis == 'frame' else batch[5]
    assert int(lens.max()) == max_len and set(lens.tolist()) == {v for v in EDGE_LENS if v <= max_len}
    ref = _oracle_step(batch, hp)
    got = _hip_step(pkg, batch, hp, 'f32')
    assert got['mel'].shape == ref['mel'].shape
    l1 = valid_mel_l1(got['mel'], ref['mel'], batch[9])
    assert l1 < 2e-5, l1
    np.testing.assert_allclose(got['mel'], ref['mel'], rtol=0, atol=3e-4)      # every element, so no single utterance hides in the mean
    assert np.abs(got['weights'] - ref['weights']).max() < 2e-4
    assert abs(got['total'] - ref['total']) <= 2e-5 * abs(ref['total'])
    rows = _grad_metrics(got['grads'], ref['grads'])
    for k, (rel, cos, _) in rows.items():
        assert rel < 3e-3, (k, rel, cos)
