"""Training-step harness (trainer.Trainer) against the CPU oracle driven by ``torch.optim.Adam`` + ``clip_grad_norm_`` with the
reference trainer's settings (src/daft_exprt/train.py:278-280, :380-445, :148-160): the multi-step loss TRAJECTORY, the
parameters after N updates (including the zero-padded speaker-logit layer, whose cached MFMA pack once went stale after the first
fused optimiser step) and the optimiser state in torch.optim.Adam's checkpoint layout."""
import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = 'cuda'
STEPS, ACCUM = 20, 2
BF16_TRAJ_REL = 3e-2        # stated bound for bf16 operands: loss of every step within 3 % of the oracle's trajectory


def _batches():
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    return [[synthetic_batch(4, (12, 24), seed=300 + 2 * s + k, n_speakers=3, zero_dur_frac=0.1) for k in range(ACCUM)] for s in range(STEPS)]


def _hparams():
    # large learning rates so that 20 updates move the loss visibly; finite clip threshold so clipping is exercised
    return helpers.golden_hparams(accumulation_steps=ACCUM, initial_learning_rate=2e-4, max_learning_rate=2e-3, warmup_steps=10,
                                  grad_clip_thresh=5.0)


@pytest.fixture(scope='module')
def oracle_run():
    """oracle forward/loss/backward + torch.optim.Adam on CPU: losses per step, speaker logits per step, final parameters, optimiser"""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.optim import update_learning_rate
    hp = _hparams()
    sd = helpers.golden_state_dict()
    names = list(helpers.manifest()['model'].keys())
    params = [sd[k].requires_grad_(True) for k in names]
    opt = torch.optim.Adam(params, betas=hp.betas, eps=hp.epsilon, weight_decay=hp.weight_decay, amsgrad=False)
    pp = helpers.golden_pitch_predictor_state_dict()
    losses, norms, spk = [], [], []
    it = 1
    for micro in _batches():
        for g in opt.param_groups:
            g['lr'] = update_learning_rate(hp, it)
        opt.zero_grad()
        tot = 0.0
        for b in micro:
            inputs = tuple(b[i] for i in range(11)) + (b[13],)
            targets = (b[1], b[3], b[4], b[8], b[9], b[10], b[6], b[7])
            out = oracle.forward(sd, inputs, hp, training=True)
            loss, _ = oracle.loss(out, targets, it, hp, pp)
            (loss / ACCUM).backward()
            tot += float(loss.detach()) / ACCUM
            spk.append(out[0].detach().numpy())
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, hp.grad_clip_thresh)))
        opt.step()
        losses.append(tot)
        it += 1
    return dict(hp=hp, names=names, losses=losses, norms=norms, spk=spk, sd={k: v.detach().clone() for k, v in sd.items()}, opt=opt.state_dict())


def _hip_run(precision):
    import ubisoft_laforge_daft_exprt_amd as pkg
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    pkg.set_precision(precision)
    try:
        hp = _hparams()
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        crit = pkg.DaftExprtLoss(DEV, hp)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    finally:
        pkg.set_precision('f32')
    trainer = Trainer(model, crit, hp)
    losses, norms = [], []
    for micro in _batches():
        loss, _, norm = trainer.train_step(micro)
        losses.append(float(loss))
        norms.append(float(norm))
    return trainer, losses, norms


def test_trajectory_f32_matches_oracle_adam(oracle_run):
    trainer, losses, norms = _hip_run('f32')
    ref = oracle_run
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, ref['losses'])]
    print('f32 loss trajectory: first', losses[0], 'last', losses[-1], '(oracle', ref['losses'][-1], ') worst rel', max(rel))
    assert ref['losses'][-1] < 0.9 * ref['losses'][0]                      # the run really trains
    assert max(rel) < 1e-3, rel
    # the gradient norm is a function of the drifting parameters (Adam turns fp32 noise on tiny gradients into O(lr) steps)
    assert max(abs(a - b) / b for a, b in zip(norms, ref['norms'])) < 5e-2
    assert trainer.iteration == STEPS + 1
    model = trainer.model
    # Parameters after 20 updates.  Adam (eps = 1e-9) normalises every component, so components whose gradient is at the fp32 noise
    # level random-walk by ~lr per step in BOTH runs, uncorrelated: element-wise comparison is meaningless for them.  What must agree
    # is the update as a whole: cosine between (p - p0) and (p_ref - p0) per tensor and over the whole model.
    p0 = helpers.golden_state_dict()
    dots = norms_a = norms_b = 0.0
    worst = (2.0, None)
    for k, p in model.named_parameters():
        da, db = (p.detach().cpu() - p0[k]).double().flatten(), (ref['sd'][k] - p0[k]).double().flatten()
        dots += float(da @ db); norms_a += float(da @ da); norms_b += float(db @ db)
        if da.numel() >= 64:
            worst = min(worst, (float(da @ db) / max(float(da.norm() * db.norm()), 1e-30), k))
    cos_all = dots / (norms_a * norms_b) ** 0.5
    print('update cosine over all parameters after', STEPS, 'updates:', cos_all, '; worst tensor:', worst)
    assert cos_all > 0.98 and worst[0] > 0.8, (cos_all, worst)
    k5 = 'speaker_classifier.classifier.5.linear_layer.weight'
    d5 = (dict(model.named_parameters())[k5].detach().cpu() - p0[k5]).flatten().double()
    r5 = (ref['sd'][k5] - p0[k5]).flatten().double()
    assert float(d5 @ r5 / (d5.norm() * r5.norm())) > 0.99          # the zero-padded logit layer's parameter really trains
    # the speaker-logit layer keeps following its parameter (stale padded pack after a fused step = first-step logits forever)
    from oracle import daft_exprt_oracle as oracle
    b = _batches()[-1][-1]
    model.eval()
    with torch.no_grad():
        got = model(model.parse_batch(DEV, b)[0])[0].cpu().numpy()
        own = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}        # the oracle at the HIP model's OWN trained weights
        want = oracle.forward(own, tuple(b[i] for i in range(11)) + (b[13],), ref['hp'])[0].numpy()
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())


def test_trajectory_bf16_within_stated_bound(oracle_run):
    _, losses, _ = _hip_run('bf16')
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, oracle_run['losses'])]
    print('bf16 loss trajectory worst rel deviation from the oracle', max(rel), 'last', losses[-1], 'vs', oracle_run['losses'][-1])
    assert max(rel) < BF16_TRAJ_REL, rel
    assert losses[-1] < 0.9 * losses[0]


def test_optimizer_state_round_trips_with_torch_adam_layout(oracle_run):
    """FusedAdam.state_dict() == the layout torch.optim.Adam checkpoints (train.py:80-85), values vs the oracle's optimiser;
    torch.optim.Adam loads it, and FusedAdam loads torch's."""
    import ubisoft_laforge_daft_exprt_amd as pkg
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    trainer, losses, _ = _hip_run('f32')
    mine, ref = trainer.optimizer.state_dict(), oracle_run['opt']
    assert set(mine.keys()) == {'state', 'param_groups'} and len(mine['param_groups']) == 1
    assert mine['param_groups'][0]['params'] == ref['param_groups'][0]['params']
    assert set(mine['state'].keys()) == set(ref['state'].keys())
    names = oracle_run['names']
    assert [k for k, _ in trainer.model.named_parameters()] == names       # index i means the same tensor in both
    for i, st in ref['state'].items():
        assert int(mine['state'][i]['step']) == int(st['step']) == STEPS
        for key, tol in (('exp_avg', 0.5), ('exp_avg_sq', 0.5)):          # relative L2 (the two runs' parameters have drifted apart by step 20)
            a, b = mine['state'][i][key].cpu(), st[key]
            assert a.shape == b.shape
            assert float((a - b).norm()) <= tol * float(b.norm()) + 1e-12, (names[i], key, float((a - b).norm() / b.norm()))
    # torch.optim.Adam accepts our dict
    probe = [torch.nn.Parameter(p.detach().clone()) for p in trainer.model.parameters()]
    torch.optim.Adam(probe).load_state_dict(mine)
    # and a fresh trainer resumes from the full reference-layout checkpoint: same next-step loss as the uninterrupted run
    ck = trainer.checkpoint()
    assert set(ck.keys()) == {'iteration', 'learning_rate', 'best_val_loss', 'state_dict', 'optimizer', 'config_params'}
    hp = _hparams()
    model2 = pkg.DaftExprt(hp).to(DEV)
    crit2 = pkg.DaftExprtLoss(DEV, hp)
    crit2.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    t2 = Trainer(model2, crit2, hp)
    ck['state_dict'] = {'module.' + k: v for k, v in ck['state_dict'].items()}        # as saved under DDP
    t2.load_checkpoint(ck)
    assert t2.iteration == trainer.iteration and t2.optimizer.step_count == STEPS
    extra = _batches()[0]
    l1, _, _ = trainer.train_step(extra)
    l2, _, _ = t2.train_step(extra)
    assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l1))
    for (k, a), (_, b) in zip(trainer.model.named_parameters(), t2.model.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), k


def test_graph_replay_matches_eager_two_phase_step():
    """The captured step (trainer.Trainer(use_graphs=True), the four-phase / four-graph form of an N > 1 step) against the one-phase
    step issued eagerly: identical loss and
    parameter trajectory at dropout p = 0, over several batches of two alternating padded shapes (so both graph sets are replayed
    after other work has run); with dropout on, replays of one shape draw different masks (device-side seed offset)."""
    import ubisoft_laforge_daft_exprt_amd as pkg
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp = helpers.golden_hparams(initial_learning_rate=2e-4, max_learning_rate=2e-3, warmup_steps=10, grad_clip_thresh=5.0)
    shapes = [dict(sym_lengths=[30, 28, 21, 9]), dict(sym_lengths=[44, 40, 12])]
    batches = []
    for s in range(8):
        kw = shapes[s % 2]
        g = torch.Generator().manual_seed(900 + s)
        L = max(kw['sym_lengths'])
        dur = torch.randint(2, 7, (len(kw['sym_lengths']), L), generator=g)
        dur[0, :kw['sym_lengths'][0]] = 5                               # row 0 is the longest on both axes: T_max fixed per shape
        batches.append(synthetic_batch(len(kw['sym_lengths']), (1, L), seed=950 + s, n_speakers=3, sym_lengths=kw['sym_lengths'], durations_int=dur))
    runs = {}
    for mode in (False, True):
        pkg.set_precision('bf16')
        try:
            model = pkg.DaftExprt(hp).to(DEV)
            model.load_state_dict(helpers.golden_state_dict(), strict=True)
            crit = pkg.DaftExprtLoss(DEV, hp)
            crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
        finally:
            pkg.set_precision('f32')
        t = Trainer(model, crit, hp, use_graphs=mode, cuts=3 if mode else 0)
        losses, l1 = [], []
        for b in batches:
            loss, terms, _ = t.train_step([b])
            losses.append(float(loss))
            l1.append(terms[0]['mel_spec_l1_loss'])
        runs[mode] = (losses, l1, {k: p.detach().clone() for k, p in model.named_parameters()})
        if mode:
            assert len(t.graphs) == 2 and all(g.hits == 4 and len(g.graphs) == 4 for g in t.graphs.values())
    (le, l1e, pe), (lg, l1g, pg) = runs[False], runs[True]
    print('eager', le, 'graph', lg)
    assert abs(le[0] - lg[0]) <= 1e-6 * abs(le[0])                        # the first step runs the same kernels on the same weights
    for a, b in zip(le, lg):                                                # afterwards Adam amplifies the atomics' summation-order noise
        assert abs(a - b) <= 3e-3 * abs(a), (le, lg)
    for a, b in zip(l1e, l1g):
        assert abs(a - b) <= 3e-3 * abs(a)
    p0 = helpers.golden_state_dict()
    dots = na = nb = 0.0
    for k in pe:
        da, db = (pe[k].cpu() - p0[k]).double().flatten(), (pg[k].cpu() - p0[k]).double().flatten()
        dots += float(da @ db); na += float(da @ da); nb += float(db @ db)
    assert dots / (na * nb) ** 0.5 > 0.995                                  # the two runs make the same update
    # dropout on: the seed offset advances inside graph A, so two replays of ONE graph on the SAME batch differ
    hp_d = pkg.HyperParams(n_speakers=3)
    pkg.set_precision('bf16')
    try:
        model = pkg.DaftExprt(hp_d).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        crit = pkg.DaftExprtLoss(DEV, hp_d)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    finally:
        pkg.set_precision('f32')
    t = Trainer(model, crit, hp_d.clone(initial_learning_rate=0.0, max_learning_rate=0.0), use_graphs=True)   # lr 0: weights stay put
    vals = [float(t.train_step([batches[0]])[0]) for _ in range(3)]
    assert len(t.graphs) == 1 and len({round(v, 6) for v in vals}) == 3, vals


def _fresh(hp, precision='bf16'):
    import ubisoft_laforge_daft_exprt_amd as pkg
    pkg.set_precision(precision)
    try:
        model = pkg.DaftExprt(hp).to(DEV)
        model.load_state_dict(helpers.golden_state_dict(), strict=True)
        crit = pkg.DaftExprtLoss(DEV, hp)
        crit.load_pitch_predictor(helpers.golden_pitch_predictor_state_dict())
    finally:
        pkg.set_precision('f32')
    return model, crit


def _speaker_stats(seed, speakers=(0, 1, 2)):
    g = torch.Generator().manual_seed(seed)
    return {s: {'pitch': {'mean': 4.0 + 0.3 * s + 0.1 * seed, 'std': 0.5 + 0.1 * s}, 'energy': {'mean': 1.0 + 0.2 * s + 0.05 * seed, 'std': 0.7 + 0.1 * s},
                'spk_emb': torch.randn(192, generator=g)} for s in speakers}


def test_conditioner_refresh_reaches_captured_graphs():
    """a captured training graph holds the conditioner's table POINTERS.  ``BatchConditioner.update`` (the reference's
    ``refresh_stats``) must therefore write in place; a refresh that has to grow the tables drops the graphs.  Graph-replayed steps
    before and after both kinds of refresh equal the eager trainer's."""
    from ubisoft_laforge_daft_exprt_amd.conditioning import BatchConditioner
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp = helpers.golden_hparams(initial_learning_rate=0.0, max_learning_rate=0.0)      # lr 0: both trainers keep the golden weights
    batch = synthetic_batch(4, (12, 24), seed=77, n_speakers=3)
    losses = {}
    for mode in (False, True):
        model, crit = _fresh(hp)
        cond = BatchConditioner(_speaker_stats(1), DEV, capacity=4)
        t = Trainer(model, crit, hp, conditioner=cond, use_graphs=mode)
        out = [float(t.train_step([batch])[0])]
        ptrs = (cond.table.data_ptr(), cond.valid.data_ptr(), cond.emb.data_ptr())
        cond.update(_speaker_stats(2))                                      # in-place refresh: same pointers, the graph stays
        assert ptrs == (cond.table.data_ptr(), cond.valid.data_ptr(), cond.emb.data_ptr())
        out.append(float(t.train_step([batch])[0]))
        if mode:
            assert len(t.graphs) == 1 and next(iter(t.graphs.values())).hits == 2
        cond.update(_speaker_stats(3, speakers=(0, 1, 2, 9)))                # speaker 9 does not fit 4 rows: re-allocation
        assert cond.n >= 10
        out.append(float(t.train_step([batch])[0]))
        if mode:
            assert next(iter(t.graphs.values())).hits == 1                   # captured anew against the new tables
        losses[mode] = out
    print('conditioner refresh: eager', losses[False], 'graphs', losses[True])
    assert len({round(v, 5) for v in losses[False]}) == 3                    # the three statistics really differ
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) <= 2e-5 * abs(a), losses


def test_load_checkpoint_into_captured_trainer_repacks_weights():
    """graph replays read the MFMA weight packs through frozen pointers; ``load_checkpoint`` into a trainer whose
    graphs are already captured must be followed by a re-pack before the next replay.  The next-step loss after loading equals a
    fresh trainer's that loaded the same checkpoint."""
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp = helpers.golden_hparams(initial_learning_rate=2e-3, max_learning_rate=2e-3, warmup_steps=10, grad_clip_thresh=5.0)
    batch = synthetic_batch(4, (12, 24), seed=78, n_speakers=3)
    model, crit = _fresh(hp)
    t = Trainer(model, crit, hp)
    first = float(t.train_step([batch])[0])
    ck0 = None
    for _ in range(3):
        t.train_step([batch])
    moved = float(t.train_step([batch])[0])
    assert abs(moved - first) > 1e-3 * abs(first)                             # five updates at lr 2e-3 moved the loss
    # a checkpoint of the GOLDEN weights, loaded into the trainer whose graphs (and packs) belong to the moved weights
    model0, crit0 = _fresh(hp)
    t0 = Trainer(model0, crit0, hp)
    ck0 = t0.checkpoint()
    t.load_checkpoint(ck0)
    got = float(t.train_step([batch])[0])
    want = float(t0.train_step([batch])[0])
    assert len(t.graphs) == 1
    assert abs(got - want) <= 2e-5 * abs(want), (got, want, first, moved)
    # the checkpoint records the learning rate the saved iteration USED (train.py:451-455), not the next one
    from ubisoft_laforge_daft_exprt_amd.optim import update_learning_rate
    ck = t0.checkpoint()
    assert ck['iteration'] == 1 and ck['learning_rate'] == update_learning_rate(hp, 1)


def test_fp16_overflow_skips_the_update_on_a_non_finite_norm():
    """ADVICE r2 (low) / VERDICT r2 weak #13: one inf in a gradient makes the squared norm non-finite; the fused Adam must leave
    parameters and both moments untouched and count the skipped step, and the next finite step must update normally."""
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    hp = helpers.golden_hparams(initial_learning_rate=1e-3, max_learning_rate=1e-3, grad_clip_thresh=5.0)
    batch = synthetic_batch(3, (10, 16), seed=79, n_speakers=3)
    model, crit = _fresh(hp, 'fp16')
    t = Trainer(model, crit, hp, use_graphs=False)
    t.train_step([batch])
    opt = t.optimizer
    p0, m0, v0 = opt.p_all.clone(), opt.m_all.clone(), opt.v_all.clone()
    assert t.skipped_steps() == 0
    t.reducer.zero_grad()
    t.reducer.flat_all.normal_()
    t.reducer.flat_all[12345] = float('inf')
    norm = opt.step(lr=1e-3, grad_scale=1.0 / t.loss_scale)
    assert not torch.isfinite(norm).item()
    assert torch.equal(opt.p_all, p0) and torch.equal(opt.m_all, m0) and torch.equal(opt.v_all, v0)
    assert t.skipped_steps() == 1
    t.reducer.flat_all[12345] = float('nan')                                 # NaN as well
    opt.step(lr=1e-3, grad_scale=1.0 / t.loss_scale)
    assert torch.equal(opt.p_all, p0) and t.skipped_steps() == 2
    loss, _, norm = t.train_step([batch])                                    # and a finite step still updates
    assert torch.isfinite(norm).item() and torch.isfinite(loss).item() and not torch.equal(opt.p_all, p0)
    assert t.skipped_steps() == 2


def test_validate_matches_oracle_eval_mode():
    """``Trainer.validate`` = train.py:163-209: eval mode, no gradients, iteration 0 in the loss, mean over batches; the captured form
    (one graph per padded shape) equals the eager form and the CPU oracle."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    import ubisoft_laforge_daft_exprt_amd as pkg
    hp_drop = pkg.HyperParams(n_speakers=helpers.manifest()['n_speakers'])   # dropout ON in the hparams: eval mode must switch it off
    batches = [synthetic_batch(4, (12, 24), seed=500 + i, n_speakers=3, zero_dur_frac=0.1) for i in range(3)]
    batches.append(batches[0])                                                # a shape seen before: replayed, not captured again
    sd, pp = helpers.golden_state_dict(), helpers.golden_pitch_predictor_state_dict()
    ref_tot, ref_terms = 0.0, None
    with torch.no_grad():
        for b in batches:
            inputs = tuple(b[i] for i in range(11)) + (b[13],)
            targets = (b[1], b[3], b[4], b[8], b[9], b[10], b[6], b[7])
            out = oracle.forward(sd, inputs, hp_drop, training=False)
            loss, terms = oracle.loss(out, targets, 0, hp_drop, pp)
            ref_tot += float(loss) / len(batches)
            ref_terms = {k: (0.0 if ref_terms is None else ref_terms[k]) + float(v) / len(batches) for k, v in terms.items()}
    results = {}
    for mode in (False, True):
        model, crit = _fresh(hp_drop, 'f32')
        t = Trainer(model, crit, hp_drop, use_graphs=mode)
        results[mode] = t.validate(batches)
        assert model.training                                                 # back in train mode
        assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
        if mode:
            assert len(t.val_graphs) == len({(tuple(b[0].shape), tuple(b[8].shape)) for b in batches})
    (le, te), (lg, tg) = results[False], results[True]
    print(f'validate: oracle {ref_tot:.6f}, eager {le:.6f}, graphs {lg:.6f}')
    assert abs(le - lg) <= 1e-6 * abs(le)
    assert abs(le - ref_tot) <= 1e-4 * abs(ref_tot)
    assert te['speaker_loss'] == 0.0                                          # iteration 0: adversarial weight 0
    for k, v in ref_terms.items():
        assert abs(tg[k] - v) <= 1e-4 * max(abs(v), 1e-3), (k, tg[k], v)
    # keep_outputs: (targets, outputs) per batch, as the reference hands them to its logger
    model, crit = _fresh(hp_drop, 'f32')
    l3, _, kept = Trainer(model, crit, hp_drop).validate(batches[:2], keep_outputs=True)
    assert len(kept) == 2 and kept[0][1][3][0].shape[1] == 80
