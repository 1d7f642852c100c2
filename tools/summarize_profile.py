"""Turns rocprofv3 CSV output into the summaries kept under profiles/.

  kernel stats :  python tools/summarize_profile.py stats <kernel-trace dir> <steps_in_run> <out.csv> [bench kernel table .json] [pmc .json]
                  per rocprof kernel: calls / step, average duration, share of the step; joined (by kernel family) with the
                  algorithmic FLOPs / bytes of bench.py --kernel-table, so every memory-bound kernel gets
                  algorithmic bytes / rocprof average duration / 8 TB/s, and with the PMC traffic next to it
  PMC traffic  :  python tools/summarize_profile.py pmc <FETCH_SIZE dir> <WRITE_SIZE dir> <precision> <out.json>
                  HBM bytes per launch per kernel family from the two counter passes (separate --pmc runs, as
                  MI355X_MICROARCH.md prescribes: FETCH_SIZE is reported at half the bytes of wide coalesced reads on gfx950 -> doubled;
                  both counters are in KB)
"""
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict

# bench.py / profiling.py kernel label -> regex on the (demangled) rocprof kernel name.  A label may cover several device kernels
# (the attention backward is a dQ and a dK/dV launch per call): their per-call figures add up.
LABELS = [
    ('ff_pair_kernel<fwd>', r'ff_pair_kernel<false, true'), ('ff_pair_kernel<bwd>', r'ff_pair_kernel<true, false'),
    ('conv_dk_kernel<3>', r'conv_dk_kernel<3,'), ('conv_dk_kernel<1>', r'conv_dk_kernel<1,'),
    ('conv_ws_kernel<3>', r'conv_ws_kernel<3,'), ('conv_ws_kernel<1>', r'conv_ws_kernel<1,'),
    ('conv_gemm_kernel<bf16>', r'conv_gemm_kernel<(__bf16|__hip_bfloat16|DF16b)|conv_gemm_kernelIDF16|conv_gemm_kernel<bool'), ('conv_gemm_kernel<f32>', r'conv_gemm_kernel<float'),
    ('wgrad_bf16_kernel<3,wide>', r'wgrad_bf16_kernel<3, (true|false), (true|false), 4>'), ('wgrad_bf16_kernel<1,wide>', r'wgrad_bf16_kernel<1, (true|false), (true|false), 4>'),
    ('wgrad_bf16_kernel<3>', r'wgrad_bf16_kernel<3,'), ('wgrad_bf16_kernel<1>', r'wgrad_bf16_kernel<1,'),
    ('wgrad_kernel<3>', r'wgrad_kernel<3>'), ('wgrad_kernel<1>', r'wgrad_kernel<1>'),
    ('attn_fwd+proj_ln', r'attn_proj_ln_fwd'), ('attn_fwd', r'attn_fwd'), ('attn_bwd (dq + dkv)', r'attn_bwd_(dq|dkv)|attn_delta'),
    ('ln_fwd_kernel<128>', r'ln_fwd_kernel<128'), ('ln_fwd_kernel<1024>', r'ln_fwd_kernel<1024|ln_fwd_kernelILi1024E'),
    ('ln_bwd_kernel<128>', r'ln_bwd_kernel<128'), ('ln_bwd_kernel<1024>', r'ln_bwd_kernel<1024|ln_bwd_kernelILi1024E'), ('loss_finalize', r'loss_finalize_kernel'), ('proj_ln_fwd', r'proj_ln_fwd_kernel'),
    ('upsample_fwd', r'upsample_fwd_kernel'), ('upsample_bwd (dsigma + dxs)', r'upsample_(bwd|dxs)_kernel'),
    ('upsample_prep', r'upsample_prep_kernel'), ('upsample_sym_bwd', r'upsample_sym_bwd_kernel'),
    ('adam_kernel', r'adam_kernel'), ('sumsq_kernel', r'sumsq_kernel'), ('mel_stats', r'mel_stats_kernel'), ('mel_grad', r'mel_grad_kernel'),
    ('add_pos', r'add_pos_kernel'), ('accent_sum', r'accent_sum_kernel'), ('scalar_conv_wgrad', r'scalar_conv_wgrad_kernel'),
    ('transpose', r'transpose_kernel'), ('mask_rows', r'mask_rows_kernel'), ('mean_pool', r'mean_pool_kernel'),
    ('mean_pool_bwd', r'mean_pool_bwd_kernel'), ('channel_affine', r'channel_affine_kernel'), ('relu_bwd', r'relu_bwd_kernel'),
    ('colsum', r'colsum_kernel'), ('pack_weights_batched', r'pack_weights_batched|pack_weights_flat'),
    ('pitch_chain<fwd>', r'pitch_fwd_kernel'), ('pitch_chain<bwd>', r'pitch_bwd_kernel'),
]


def _find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    if not hits:
        raise SystemExit(f'no *{suffix} under {d}')
    return hits[0]


_DEMANGLED = {}


def demangle(name):
    if name not in _DEMANGLED:
        out = name
        if name.startswith('_Z'):
            try:
                tool = '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'           # knows the bf16 / fp16 manglings (DF16b, DF16_) that binutils' c++filt may not
                tool = tool if os.path.exists(tool) else 'c++filt'
                out = subprocess.run([tool, name], capture_output=True, text=True, timeout=10).stdout.strip() or name
            except (OSError, subprocess.SubprocessError):
                pass
        _DEMANGLED[name] = re.sub(r'\(anonymous namespace\)::', '', out)
    return _DEMANGLED[name]


def label_of(kernel_name):
    for label, pat in LABELS:
        if re.search(pat, kernel_name):
            return label
    return None


def stats(d, steps, out, table_json=None, pmc_json=None):
    """``steps`` is only a fallback.  The trace is cut to WHOLE STEPS: everything after the first ``adam_kernel`` launch up to and including the
    last one (a step ends with the fused Adam + the one-launch re-pack that follows it, which is attributed to the next window: the same count
    per step either way).  Round 2 divided the whole trace by the step count, so the ~1,600 device-to-device copies of model SETUP
    (load_state_dict, optimiser re-homing) appeared as "62 copyBuffer launches per step"."""
    with open(_find(d, 'kernel_trace.csv')) as f:
        trace = sorted(csv.DictReader(f), key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(trace) if 'adam_kernel' in r['Kernel_Name']]
    if len(adam) >= 2:
        trace, steps = trace[adam[0] + 1:adam[-1] + 1], len(adam) - 1
    rows = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in trace:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e = rows[demangle(r['Kernel_Name'])]
        e[0] += 1; e[1] += dur; e[2] = min(e[2], dur); e[3] = max(e[3], dur)
    if trace:
        span = (int(trace[-1]['End_Timestamp']) - int(trace[0]['Start_Timestamp'])) / 1e6
        busy = sum(e[1] for e in rows.values()) / 1e3
        print('steady-state window: %d steps, %.3f ms per step wall, %.3f ms per step inside kernels (%.1f %% of the wall: the rest is gaps between launches)'
              % (steps, span / steps, busy / steps, 100 * busy / span))
    total = sum(e[1] for e in rows.values())
    table = json.load(open(table_json))['kernels'] if table_json else {}
    precision = json.load(open(table_json))['precision'] if table_json else 'bf16'
    pmc = json.load(open(pmc_json)).get(precision, {}) if pmc_json and os.path.exists(pmc_json) else {}
    # per label: rocprof time and launches per step (a label may cover several device kernels)
    lab_time, lab_calls = defaultdict(float), defaultdict(float)
    for k, e in rows.items():
        lab = label_of(k)
        if lab:
            lab_time[lab] += e[1] / steps
            lab_calls[lab] += e[0] / steps
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls_per_step', 'ms_per_step', 'avg_us', 'min_us', 'max_us', 'percent', 'family', 'bound',
                    'family_algorithmic_per_call', 'family_achieved_from_rocprof', 'unit', 'frac_of_peak', 'family_pmc_hbm_bytes_per_call', 'pmc_over_algorithmic'])
        for k, e in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            lab = label_of(k)
            extra = [lab or '', '', '', '', '', '', '', '']
            t = table.get(lab) if lab else None
            if t and 'achieved' in t:
                calls = t['launches']                                  # calls of the C entry point per step (bench's own count)
                us_per_call = lab_time[lab] / calls
                if t['bound'] == 'mfma':
                    work = t['achieved'] * 1e12 * t['avg_us'] * 1e-6   # algorithmic FLOPs per call
                    ach, unit, peak = work / (us_per_call * 1e-6) / 1e12, 'TFLOP/s', t['peak']
                else:
                    work = t['algorithmic_bytes_per_launch']
                    ach, unit, peak = work / (us_per_call * 1e-6) / 1e9, 'GB/s', t['peak']
                traffic = pmc.get(lab, {}).get('hbm_bytes_per_launch')
                alg_b = t.get('algorithmic_bytes_per_launch')
                extra = [lab, t['bound'], int(work), round(ach, 1), unit, round(ach / peak, 4), traffic or '',
                         round(traffic / alg_b, 2) if traffic and alg_b else '']
            w.writerow([k, round(e[0] / steps, 2), round(e[1] / 1e3 / steps, 4), round(e[1] / e[0], 2), round(e[2], 2), round(e[3], 2),
                        round(100 * e[1] / total, 2)] + extra)
        w.writerow(['TOTAL', '', round(total / 1e3 / steps, 4), '', '', '', 100.0] + [''] * 8)
    print('kernel time per step: %.3f ms over %d kernels' % (total / 1e3 / steps, len(rows)))


def _counter(d, name):
    acc = defaultdict(lambda: [0, 0.0])
    with open(_find(d, 'counter_collection.csv')) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != name:
                continue
            lab = label_of(demangle(r['Kernel_Name']))
            if lab:
                acc[lab][0] += 1
                acc[lab][1] += float(r['Counter_Value'])
    return acc


def pmc(fetch_dir, write_dir, precision, out, calls_json=None):
    fe, wr = _counter(fetch_dir, 'FETCH_SIZE'), _counter(write_dir, 'WRITE_SIZE')
    calls = {k: v['launches'] for k, v in json.load(open(calls_json))['kernels'].items()} if calls_json else {}
    fams = {}
    for lab in fe:
        n_dev = fe[lab][0]
        fetch = fe[lab][1] * 1024.0          # KB -> bytes, summed over all profiled device launches
        write = wr[lab][1] * 1024.0 * (fe[lab][0] / max(wr[lab][0], 1))
        fams[lab] = {'device_launches_profiled': n_dev, 'fetch_size_bytes_per_device_launch_raw': int(fetch / n_dev),
                     'write_size_bytes_per_device_launch': int(write / n_dev)}
    doc = {}
    if os.path.exists(out):
        with open(out) as f:
            doc = json.load(f)
    doc['note'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over bench.py (C2 workload); values are KB in the CSV; '
                   'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); hbm_bytes_per_launch is per '
                   'call of the C entry point: the device launches of one call (e.g. attention backward = dQ + dK/dV kernels) are added up')
    # per call of the entry point = per device launch x device launches per call (ratio of profiled device launches to bench calls)
    steps = None
    for lab, v in fams.items():
        per_dev = 2 * v['fetch_size_bytes_per_device_launch_raw'] + v['write_size_bytes_per_device_launch']
        dev_per_call = 2 if lab.startswith('attn_bwd') or lab.startswith('upsample_bwd') else 1
        v['hbm_bytes_per_launch'] = int(per_dev * dev_per_call)
    doc[precision] = fams
    with open(out, 'w') as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in fams.items()}, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4], *(sys.argv[5:7]))
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
