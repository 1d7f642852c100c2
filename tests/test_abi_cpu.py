"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the header declares, and the
Python module owns the reference's state-dict layout.  No compute call is made (there is no GPU here)."""
import ctypes
import os

import pytest
import torch

from tests import helpers


def test_library_exports_every_declared_symbol():
    from ubisoft_laforge_daft_exprt_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 36
    assert os.path.exists(_lib.LIB_PATH), 'build it: python -m ubisoft_laforge_daft_exprt_amd.build'
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    lib = _lib.lib()
    assert lib.dx_version() == 1


def test_argument_validation_without_gpu():
    from ubisoft_laforge_daft_exprt_amd._lib import lib, DxError
    with pytest.raises(DxError, match='taps'):
        lib().dx_pack_weights(1, 1, None, 8, 8, 2, 0, None)           # argument checks run before any launch
    with pytest.raises(DxError, match='null'):
        lib().dx_conv_gemm(None, 0, None, None, None, 0, 1, 1, 4, 4, 1, 0, 0, None, None, None, 0, 0, None, 0, 1.0, -1, 0, 0, 0, None, None)


def test_state_dict_layout_matches_reference_manifest():
    import ubisoft_laforge_daft_exprt_amd as pkg
    man = helpers.manifest()
    model = pkg.DaftExprt(helpers.golden_hparams())
    sd = model.state_dict()
    assert list(sorted(sd)) == list(sorted(man['model']))
    for k, shape in man['model'].items():
        assert list(sd[k].shape) == shape, k
        assert sd[k].dtype == torch.float32
    assert sum(p.numel() for p in model.parameters()) == man['n_parameters']
    assert len(list(model.buffers())) == 0
    model.load_state_dict(helpers.golden_state_dict(), strict=True)
    # DDP-style 'module.' prefixed checkpoints are stripped by the reference's consumers (fine_tune.py:40); same keys here
    nopm = pkg.DaftExprt(helpers.golden_hparams(post_mult_weight=0.0))
    assert 'style_adapter.post_multipliers' not in nopm.state_dict()
    assert nopm.style_adapter.post_multipliers == 1.0


def test_no_cpu_fallback_and_reference_errors():
    import ubisoft_laforge_daft_exprt_amd as pkg
    model = pkg.DaftExprt(helpers.golden_hparams())
    inputs, _ = helpers.case_inputs(helpers.load_case('train_single'))
    with pytest.raises(ValueError):
        model(inputs[:11])
    with pytest.raises(RuntimeError, match='GPU'):
        model(inputs)
    with pytest.raises(ValueError):
        model.parse_batch('cpu', inputs)


def test_duration_rounding_is_bit_exact_against_reference_kats():
    import json
    from ubisoft_laforge_daft_exprt_amd.durations import duration_to_integer, get_int_durations
    with open(os.path.join(helpers.GOLDEN, 'duration_kats.json')) as f:
        kats = json.load(f)
    hp = helpers.golden_hparams()
    for kat in kats['duration_to_integer']:
        hpk = hp.clone(centered=True) if kat.get('centered') else hp
        try:
            got = duration_to_integer([list(s) for s in kat['spans']], hpk)
        except (IndexError, ValueError) as exc:
            got = type(exc).__name__
        assert got == kat['expected'], kat
    g = kats['get_int_durations']
    f_out, i_out = get_int_durations(torch.tensor(g['input'], dtype=torch.float32), hp)
    assert i_out.tolist() == g['int_out'] and f_out.tolist() == g['float_out']


def test_duration_rounding_matches_oracle_on_random_rows():
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.durations import get_int_durations
    hp = helpers.golden_hparams()
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        rows = (0.2 * torch.rand(3, 40, generator=g)).float()
        rows[:, 0] = 0.1
        rows = rows * (torch.rand(3, 40, generator=g) > 0.2)
        rows[:, 0] = 0.1
        a = get_int_durations(rows.clone(), hp)
        b = oracle.get_int_durations(rows.clone(), hp)
        assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])


@pytest.mark.parametrize('centered', [False, True])
def test_host_library_durations_equal_oracle_including_error_rows(centered):
    """dx_int_durations (C, host) against the oracle's restatement of duration_to_integer (extract_features.py:69-125): bit-exact integers
    and totals over many scales and sparsities, both window conventions; utterances the reference raises on raise the same exception."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.durations import get_int_durations
    hp = helpers.golden_hparams().clone(centered=centered)
    g = torch.Generator().manual_seed(3)
    ok = raised = 0
    for trial in range(60):
        L = int(torch.randint(1, 60, (1,), generator=g))
        rows = ((0.01 + 0.3 * torch.rand(4, L, generator=g)) * (0.15 + 0.05 * trial)).float()
        rows = rows * (torch.rand(4, L, generator=g) > 0.25 * (trial % 3))
        try:
            want = oracle.get_int_durations(rows.clone(), hp)
        except Exception as exc:                     # noqa: BLE001 -- the exception TYPE is the expected value here
            with pytest.raises(type(exc)):
                get_int_durations(rows.clone(), hp)
            raised += 1
            continue
        f, i, totals = get_int_durations(rows.clone(), hp, return_totals=True)
        assert torch.equal(i, want[1]) and torch.equal(f, want[0]) and totals == want[1].sum(dim=1).tolist()
        ok += 1
    assert ok >= 20 and raised >= 3, (ok, raised)


def test_synthetic_batch_contract():
    from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch
    batch = synthetic_batch(**CONFIGS['C1'])
    assert len(batch) == 14
    symbols, dur_f, dur_i, s_e, s_p, in_l, f_e, f_p, mel, out_l, spk, _, _, emb = batch
    assert torch.equal(dur_i.sum(1), out_l) and in_l.tolist() == sorted(in_l.tolist(), reverse=True)
    assert mel.shape == (4, 80, int(out_l.max())) and emb.shape == (4, 192)
    for b in range(4):
        assert (symbols[b, int(in_l[b]):] == 0).all() and (mel[b, :, int(out_l[b]):] == 0).all()


def test_feature_reader_and_collate_vs_reference_golden(tmp_path):
    """f-4: the on-disk format reader + collator reproduce DaftExprtDataLoader/DaftExprtDataCollate bit for bit."""
    import numpy as np
    from ubisoft_laforge_daft_exprt_amd.features import SYMBOLS_ENGLISH, FeatureSet, collate, read_utterance
    assert len(SYMBOLS_ENGLISH) == 76
    case = helpers.load_case('feature_collate')
    list_file, rows = helpers.write_synthetic_features(str(tmp_path))
    hp = helpers.golden_hparams(stats=helpers.FEATURE_STATS)
    for raw, tag in ((False, 'norm'), (True, 'raw')):
        out = collate([read_utterance(d, f, s, hp, return_raw_stats=raw) for d, f, s in rows], hp, pin_memory=False)
        assert len(out) == 14
        for k, t in enumerate(out):
            if torch.is_tensor(t):
                ref = case[f'{tag}/{k}']
                assert t.dtype == torch.from_numpy(ref).dtype and np.array_equal(t.numpy(), ref), (tag, k)
        assert list(out[12]) == list(case[f'{tag}/files'])
    shard = FeatureSet(list_file, hp, batch_size=2, rank=1, world=2)
    assert len(shard) == 1 and [f for b in shard for f in b[12]] != []
    with open(os.path.join(str(tmp_path), 'utt000.frames_f0'), 'a') as f:
        f.write('1.0\n')                                     # corrupt: one frame too many
    with pytest.raises(ValueError, match='frames_pitch'):
        read_utterance(*rows[0], hp)


def test_every_launching_entry_point_can_be_priced():
    """profiling.price() (bench.py's per-kernel roofline accounting) accepts the argument list of every C-ABI launch function."""
    from ubisoft_laforge_daft_exprt_amd import profiling
    from ubisoft_laforge_daft_exprt_amd._lib import parse_header
    geom = profiling.Geometry([[5, 9, 12], [3, 4]])
    for name, (_, _, argnames) in parse_header(with_names=True).items():
        if 'stream' not in argnames:
            continue
        args = {k: 8 for k in argnames}
        args.update(B=3, N=12, bf16=1)
        label, bound, flops, byt = profiling.price(name, args, geom)
        assert isinstance(label, str) and bound in ('mfma', 'hbm')
        assert flops is None or flops > 0
        assert byt is None or byt >= 0
    # valid rows, not padded rows, are credited
    a = dict(B=3, N=12, Cin=128, Cout=128, taps=3, bf16=1, x_bf16=0, y_bf16=0, aux_bf16=0, accumulate=0, relu_aux=0, lens=1)
    assert profiling.price('dx_conv_gemm', a, geom)[2] == 2.0 * 3 * 128 * 128 * (5 + 9 + 12)


def test_gradient_kernels_use_no_packed_fma_that_reads_the_high_half_into_the_low_result():
    """Round 2's "lost update" was bisected (round 3, tools/experiment_fork_wgrad.py) to ``v_pk_fma_f32 ... op_sel:[0,1,0]`` -- the form
    hipcc chose for the middle tap of ``scalar_conv_wgrad_kernel`` -- returning wrong low results while an MFMA kernel of another
    stream shared the CUs.  The kernel now spells its tap products as single ``v_fma_f32`` instructions; this test compiles the source
    to gfx950 assembly and checks that no packed FMA of that form is left anywhere in the file that holds the row / gradient kernels."""
    import os, shutil, subprocess, tempfile
    if shutil.which('hipcc') is None:
        import pytest
        pytest.skip('hipcc not on PATH')
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ubisoft_laforge_daft_exprt_amd', 'csrc')
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'dx_rows.s')
        subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-S', '--cuda-device-only', '-o', out, os.path.join(csrc, 'dx_rows.hip')],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad = [ln.strip() for ln in text.splitlines() if 'v_pk_fma_f32' in ln and 'op_sel:[0,1' in ln]
    assert not bad, bad[:4]
    assert 'scalar_conv_wgrad_kernel' in text
