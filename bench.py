"""Headline benchmark: valid mel frames / s of one Daft-Exprt training step (forward + loss + backward) on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): 48 utterances per GPU, 50-120 symbols, 2-12 frames per symbol
(T_max ~ 840), synthetic data, seeded random-init weights, dropout ON, energy- and pitch-consistency losses ON with a
seeded frozen pitch predictor.  The timed step is the whole training step: forward + loss + backward (+ the bucketed RCCL
gradient all-reduce when N > 1) + the fused Adam update and the one-launch weight re-pack that follows it (the metric is
"fwd+bwd"; the optimiser is a superset of that work, ~1.5 % of a step, and stays inside the timed region so that no work of
a real training step is skipped).  ``python bench.py --gpus N`` starts its N ranks itself.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

def cpu_baseline(hp, n_threads, config, n_speakers):
    """The CPU oracle (a port that keeps the reference's op structure, so its cost is the reference's cost: BASELINE.md) on the
    SAME synthetic batch as the GPU number: 1 warm-up + up to 3 timed steps of forward + loss (pitch predictor included) +
    backward, dropout on.  The bounded sample: timing stops early once ~30 s of timed CPU work have been spent."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.loss import pitch_predictor_shapes
    from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict
    import ubisoft_laforge_daft_exprt_amd as pkg
    torch.set_num_threads(n_threads)
    shapes = {k: tuple(v.shape) for k, v in pkg.DaftExprt(hp.clone()).state_dict().items()}
    sd = synthetic_state_dict(shapes, 1234)
    pp_sd = synthetic_state_dict(pitch_predictor_shapes(), 1235)
    for v in sd.values():
        v.requires_grad_(True)
    cfg = dict(CONFIGS[config])
    cfg['n_speakers'] = n_speakers
    batch = synthetic_batch(**cfg)
    inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    targets = (batch[1], batch[3], batch[4], batch[8], batch[9], batch[10], batch[6], batch[7])
    frames = int(batch[9].sum())
    times = []
    for it in range(4):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = oracle.forward(sd, inputs, hp, training=True)
        total, _ = oracle.loss(out, targets, 1000, hp, pp_sd)
        total.backward()
        times.append(time.perf_counter() - t0)
        print(f'[cpu_baseline] step {it}: {times[-1]:.2f} s on {n_threads} threads', file=sys.stderr, flush=True)
        if it >= 1 and sum(times[1:]) > 30.0:                             # keep the default run within minutes
            break
    timed = times[1:] if len(times) > 1 else times
    step = sum(timed) / len(timed)
    return {'value': frames / step, 'unit': 'mel frames/s', 'cores': n_threads, 'kind': 'port',
            'sample': f'{config} batch, the same one the GPU number is on (B={cfg["batch_size"]}, {frames} valid frames), fwd+loss+bwd, '
                      f'dropout on, pitch predictor on, 1 warm-up + {len(timed)} timed steps, {step:.2f} s/step'}


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N ranks as fresh child processes (one per GPU, RCCL over xGMI)
    BEFORE anything in this process touches the GPU, relay their output, exit with their code.  The reference spawns its ranks
    itself too (train.py:548-657, mp.spawn)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--precision', default='bf16', choices=['f32', 'bf16', 'fp16'])
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C5'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--launch-dump', default='', help='write one JSON line per launch of the last (eager, event-bracketed) timed step')
    ap.add_argument('--kernel-table', default='', help='write the per-kernel roofline table of the last timed step to this JSON file')
    ap.add_argument('--no-graph', action='store_true', help='issue every launch from Python instead of replaying the two captured HIP graphs')
    ap.add_argument('--no-dropout', action='store_true', help='diagnostic only: the headline metric is measured with dropout on')
    ap.add_argument('--cuts', default='auto', help="backward phases of the trainer: 'auto' (3 cuts when N > 1, none on one GPU) or 0..3")
    ap.add_argument('--no-extras', action='store_true', help='skip the sustained run, the f32-mode step and the C4 inference batch that follow the timed region')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # ONE JSON line on stdout, nothing else: librccl prints a version banner to stdout when the process group comes up (seen in the one-rank
    # rehearsal), so from here on file descriptor 1 is stderr for every library in this process and the result goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    # Rehearsal of the multi-rank path on a ONE-GPU box (DX_BENCH_REHEARSAL=1): every rank uses cuda:0 and the gradients travel over gloo.
    # It exercises the trainer / graph / bucket-group code of N > 1 end to end; its numbers mean nothing (one GPU shared, host-staged all-reduce).
    rehearsal = os.environ.get('DX_BENCH_REHEARSAL', '0') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world == 1 and os.environ.get('DX_FORCE_COLLECTIVES', '0') == '1':
        # one-rank RCCL group on the one GPU: the N > 1 code path (four phases, four graphs, an RCCL all-reduce launched between the replays)
        # end to end with the real backend; there is nobody to exchange with, so the collectives move no data between GPUs
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import ubisoft_laforge_daft_exprt_amd as pkg
    from ubisoft_laforge_daft_exprt_amd import ops
    from ubisoft_laforge_daft_exprt_amd import _lib
    from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict

    pkg.set_precision(args.precision)
    n_speakers = 12 if world > 1 else 2                                  # C3: 11 speakers + 1; C2: single speaker + 1
    hp = pkg.HyperParams(n_speakers=n_speakers)
    if args.no_dropout:
        hp = hp.without_dropout()
    model = pkg.DaftExprt(hp).to(dev)
    model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234), strict=True)
    model.train()
    crit = pkg.DaftExprtLoss(dev, hp)
    from ubisoft_laforge_daft_exprt_amd.loss import pitch_predictor_shapes
    crit.load_pitch_predictor(synthetic_state_dict(pitch_predictor_shapes(), 1235))
    cfg = dict(CONFIGS[args.config])
    cfg['seed'] = cfg['seed'] + 1000 * rank                              # every rank owns different utterances
    cfg['n_speakers'] = n_speakers
    batch = synthetic_batch(**cfg)
    frames = int(batch[9].sum())
    symbols = int(batch[5].sum())
    pkg.manual_seed(1234 + rank)
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    # the batch is resident in HBM before the timed region starts (the reference's 14-tuple, device tensors; raw frame prosody
    # rides along in inputs[6], inputs[7] as train.py:405-419 keeps it for the consistency losses)
    dev_batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in batch)
    for i in (5, 9):
        dev_batch[i]._dx_host_lengths = batch[i].tolist()
    trainer = Trainer(model, crit, hp, use_graphs=not args.no_graph, cuts=args.cuts if args.cuts == 'auto' else int(args.cuts))
    dev_batch = trainer.resident_batch(dev_batch)      # the graphs' static input buffers ARE the resident batch
    ev_bwd = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_red = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    _orig_finish = trainer.reducer.finish

    def step(it, timed_idx=None):
        # Trainer.train_step: per backward phase [one captured graph] -> that phase's all-reduce group (async, RCCL's stream), the next
        # phase running beside it; one phase on one GPU, four when N > 1 -> wait -> fused Adam (clip, LR schedule) -> one-launch re-pack
        if timed_idx is not None:
            def finish():
                ev_bwd[timed_idx].record()
                _orig_finish()                    # waits for the bucketed all-reduces
                ev_red[timed_idx].record()
            trainer.reducer.finish = finish
        total, _terms, _norm = trainer.train_step([dev_batch])
        trainer.reducer.finish = _orig_finish
        return total

    for it in range(args.warmup):
        step(it)
    torch.cuda.synchronize()
    use_events = not args.no_kernel_events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    records = []
    host_s = 0.0                                  # time the host spends ENQUEUEING a step (no sync inside): must stay below ms_per_step
    for it in range(args.steps):
        if use_events and it == args.steps - 1:
            # Per-kernel events need one host call per launch, so the LAST timed step is issued eagerly (not as graph replays) with
            # every launch bracketed: ~25 % slower than a replayed step, 1 step in args.steps, inside the timed region.
            trainer.use_graphs = False
            _lib.set_timer(records)
        th = time.perf_counter()
        last = step(args.warmup + it, it)
        host_s += time.perf_counter() - th
    _lib.set_timer(None)
    trainer.use_graphs = not args.no_graph
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, float(frames)], device=dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_frames = float(tmax[0]), float(t[1])
    else:
        total_frames = float(frames)
    assert torch.isfinite(last).item(), 'non-finite loss in the timed region'
    exposed_ms = sum(a.elapsed_time(b) for a, b in zip(ev_bwd, ev_red)) / args.steps   # compute-stream time spent waiting in reducer.finish()
    if world > 1:
        te = torch.tensor([exposed_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        exposed_ms = float(te[0])

    roofline = roofline_hbm = table = None
    if use_events and rank == 0:
        from ubisoft_laforge_daft_exprt_amd import profiling
        geom = profiling.Geometry([batch[9].tolist(), batch[5].tolist()])
        table = profiling.summarize(records, geom, args.precision)
        step_us = 1e6 * elapsed / args.steps
        mfma = [(k, v) for k, v in table.items() if v['bound'] == 'mfma']
        if mfma:
            k, v = mfma[0]                                            # the single kernel with the most time in the step
            traffic = traffic_source = None
            for pmc_name in ('r03_pmc_traffic.json', 'r02_pmc_traffic.json'):      # PMC counters need rocprofv3: collected by tools/collect_profiles.sh, committed
                pmc = os.path.join(REPO, 'profiles', pmc_name)
                if os.path.exists(pmc):
                    with open(pmc) as f:
                        traffic = json.load(f).get(args.precision, {}).get(k, {}).get('hbm_bytes_per_launch')
                    if traffic is not None:
                        traffic_source = f'profiles/{pmc_name} (committed; rocprofv3 --pmc passes of this command, not measured in this run)'
                        break
            roofline = {'bound': 'mfma', 'kernel': k, 'achieved': v['achieved'], 'peak': v['peak'], 'unit': 'TFLOP/s', 'frac': v['frac'],
                        'traffic': traffic, 'traffic_source': traffic_source, 'algorithmic_bytes_per_launch': v['algorithmic_bytes_per_launch'], 'launches': v['launches'],
                        'avg_launch_us': v['avg_us'], 'share_of_step': round(v['total_us'] / step_us, 3),
                        'measured_on': 'events around every launch of this kernel in the last timed step',
                        'other_mfma_kernels': {kk: {'frac': vv['frac'], 'achieved': vv['achieved'], 'launches': vv['launches'], 'avg_us': vv['avg_us'],
                                                    'share_of_step': round(vv['total_us'] / step_us, 3)} for kk, vv in mfma[1:8]}}
        roofline_hbm = {k: {'achieved': v['achieved'], 'unit': 'GB/s', 'peak': v['peak'], 'frac': v['frac'], 'launches': v['launches'],
                            'avg_us': v['avg_us'], 'algorithmic_bytes_per_launch': v['algorithmic_bytes_per_launch'],
                            'share_of_step': round(v['total_us'] / step_us, 3)}
                        for k, v in table.items() if v['bound'] == 'hbm' and 'achieved' in v and v['total_us'] / step_us >= 0.002}
        if args.launch_dump:                                          # every launch of the eager step in issue order: entry point, shape arguments, us
            keep = ('B', 'N', 'Cin', 'Cout', 'taps', 'F', 'C', 'L', 'T', 'D', 'H', 'transpose', 'accumulate', 'njobs', 'rows', 'n')
            with open(args.launch_dump, 'w') as f:
                for name, a, e0, e1 in records:
                    f.write(json.dumps({'entry': name, 'label': profiling.price(name, a, geom)[0], 'us': round(1e3 * e0.elapsed_time(e1), 2),
                                        **{k: a[k] for k in keep if k in a and isinstance(a[k], (int, float))}}) + '\n')
        if args.kernel_table:
            with open(args.kernel_table, 'w') as f:
                json.dump({'precision': args.precision, 'config': args.config, 'ms_per_step': 1e3 * elapsed / args.steps, 'kernels': table}, f, indent=1)

    # ---- what follows the timed region: numbers that used to exist only in builder-run files -------------------------
    extras = {}
    ms_step = 1e3 * elapsed / args.steps
    if not args.no_extras:
        # (1) sustained: the same captured step replayed for >= 2 s (the timed region above is a 0.13 s burst; clocks sag under
        #     sustained MFMA load).  Every rank takes part (the step contains the collectives); the count comes from the max-over-ranks time.
        n_sus = max(args.steps, int(2000.0 / ms_step) + 1)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for it in range(n_sus):
            step(args.warmup + args.steps + it)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        sus = time.perf_counter() - ts
        extras['sustained'] = {'seconds': round(sus, 3), 'steps': n_sus, 'ms_per_step': round(1e3 * sus / n_sus, 3),
                               'value': round(total_frames * n_sus / sus, 1)}
    if not args.no_extras and world == 1 and rank == 0:
        # (2) the parity mode (exact-f32 MFMA operands: the mode that meets the 1e-4 mel-L1 bar), same batch, same trainer structure
        if args.precision != 'f32':
            pkg.set_precision('f32')
            try:
                m32 = pkg.DaftExprt(hp).to(dev)
                m32.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in m32.state_dict().items()}, 1234), strict=True)
                c32 = pkg.DaftExprtLoss(dev, hp)
                c32.load_pitch_predictor(synthetic_state_dict(pitch_predictor_shapes(), 1235))
            finally:
                pkg.set_precision(args.precision)
            t32 = Trainer(m32, c32, hp, use_graphs=not args.no_graph, cuts=args.cuts if args.cuts == 'auto' else int(args.cuts))
            b32 = t32.resident_batch(dev_batch)
            for _ in range(2):
                t32.train_step([b32])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(8):
                l32, _, _ = t32.train_step([b32])
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - t1) / 8
            assert torch.isfinite(l32).item()
            extras['f32_mode'] = {'ms_per_step': round(1e3 * d32, 3), 'value': round(frames / d32, 1), 'steps': 8, 'warmup': 2,
                                  'note': 'exact-f32 MFMA operands everywhere: the mode whose forward meets mel L1 <= 1e-4 vs the reference'}
            del t32, m32, c32, b32
        # (3) config 4: inference, 256 sentences per batch, bucketed graph replay, inputs resident + the batched reference-recording leg
        sys.path.insert(0, os.path.join(REPO, 'tools'))
        import bench_inference
        inf = bench_inference.measure(args.precision, 8, n=10, warm=2)
        extras['inference_c4'] = {'ms_per_batch': inf['graph_replay']['ms_per_batch'], 'frames_per_s': inf['graph_replay']['frames_per_s'],
                                  'accent_leg_ms': inf['accent_encoder_leg']['batched_graph']['ms'], 'end_to_end_frames_per_s': inf['end_to_end_graph']['frames_per_s'],
                                  'B': inf['B'], 'T_max': inf['T_max'], 'valid_frames': inf['valid_frames'], 'host_prepare_ms': inf['host_prepare_ms'],
                                  'precision': args.precision, 'launch': 'bucketed hipGraph replay, inputs resident in HBM'}
    # (4) parity of the measured mode on the measured shape, from the last committed GPU test record (tests/test_configs_gpu.py)
    parity = None
    import glob
    recs = sorted(glob.glob(os.path.join(REPO, 'profiles', f'r*_parity_{args.config.lower()}_{args.precision}.json')))
    if recs:
        with open(recs[-1]) as f:
            pr = json.load(f)
        parity = {'mode': args.precision, 'mel_l1_vs_oracle': pr.get('mel_l1'), 'bar_in_test': pr.get('bar_mel_l1'),
                  'north_star_bar': 1e-4, 'meets_north_star_bar': bool(pr.get('mel_l1', 1.0) <= 1e-4),
                  'worst_grad_rel': pr.get('worst_grad_rel'), 'worst_grad_cos': pr.get('worst_grad_cos'),
                  'source': 'profiles/' + os.path.basename(recs[-1]) + ' (tests/test_configs_gpu.py on MI355X)'}

    # whole-step algorithmic MFMA rate, SURVEY.md section 8(d): F_fwd(L, T) per utterance on VALID tokens, forward + backward = 3 x forward
    def f_fwd(L, T):
        fft = lambda n: n * 1703936 + 512 * n * n
        return T * 7569408 + T * 1536 + 8 * fft(T) + 4 * fft(L) + L * (3 * 768 + 256) + L * T * (8 + 2 * 128) + T * 20480 + 600000
    step_flop = 3.0 * sum(f_fwd(int(l), int(t)) for l, t in zip(batch[5].tolist(), batch[9].tolist())) * world
    whole_step = {'algorithmic_tflop_per_step': round(step_flop / 1e12, 4), 'achieved_tflops': round(step_flop / 1e12 / (elapsed / args.steps), 1),
                  'peak_tflops': 2500.0 if args.precision != 'f32' else 157.3 * 1.0,
                  'formula': 'SURVEY.md section 8(d): 3 x sum_b F_fwd(L_b, T_b), valid tokens only (the frozen pitch predictor of the loss is extra work that is not credited)'}
    whole_step['frac'] = round(whole_step['achieved_tflops'] / (whole_step['peak_tflops'] * world), 4)

    if rank == 0:
        result = {
            'metric': 'mel frames/sec (fwd+bwd)', 'value': round(total_frames * args.steps / elapsed, 1), 'unit': 'valid mel frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'bf16': 'bf16', 'fp16': 'f16', 'f32': 'f32'}[args.precision], 'data': 'synthetic',
            'config': {'workload': f'{args.config}: LJ-shaped training step, {cfg["batch_size"]} utterances/GPU, '
                                   f'L_max={int(batch[5].max())}, T_max={int(batch[9].max())}, {frames} valid frames/GPU/step; '
                                   'forward + loss (mel L1/L2, adversarial CE, post-mult, energy + pitch consistency) + backward'
                                   + (' + bucketed RCCL gradient all-reduce overlapped with backward' if world > 1 else '')
                                   + ' + fused Adam step (global-norm clip, LR schedule) + one-launch weight re-pack'
                                   + (', dropout OFF (diagnostic)' if args.no_dropout else ', dropout on'),
                       'operands': (f'{args.precision} MFMA operands for Conv1d/Linear GEMMs and attention, fp32 accumulate; 1024-wide hidden tensors, qkv, '
                                    f'attention context and their gradients stored {args.precision}' + (', static loss scale 4096' if args.precision == 'fp16' else ''))
                                   if args.precision != 'f32' else 'exact f32 MFMA everywhere',
                       'parallelism': f'dp{world}', 'launch': 'eager (one Python call per kernel)' if args.no_graph else f'{trainer.cut_levels + 1} captured HIP graph(s) per step (one per backward phase) + eager optimiser'},
        }
        result['rccl_ranks'] = dist.get_world_size() if world > 1 else 1
        result['exposed_allreduce_ms_per_step'] = round(exposed_ms, 4)
        result['host_enqueue_ms_per_step'] = round(1e3 * host_s / args.steps, 3)
        result['whole_step'] = whole_step
        result['exchange'] = trainer.exchange_plan()     # per group: bytes, launch point; exposed_bytes = the group launched after the last backward kernel
        result.update(extras)
        if parity is not None:
            result['parity'] = parity
        if roofline is not None:
            result['roofline'] = roofline
        if roofline_hbm:
            result['roofline_hbm'] = roofline_hbm
        if world == 1 and not args.no_cpu_baseline:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            cores = min(avail, 16)                                       # a 1-GPU box is given a 16-CPU share
            result['cpu_baseline'] = cpu_baseline(hp, cores, args.config, n_speakers)
        os.write(json_fd, (json.dumps(result) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
