"""Headline benchmark: valid mel frames / s of one Daft-Exprt training step (forward + loss + backward) on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): 48 utterances per GPU, 50-120 symbols, 2-12 frames per symbol
(T_max ~ 840), synthetic data, seeded random-init weights, dropout ON, energy- and pitch-consistency losses ON with a
seeded frozen pitch predictor, weights re-packed every step (as after an optimiser update).  The optimiser itself is not
part of the metric ("fwd+bwd", BASELINE.json) and is excluded.  One JSON line is printed by rank 0.
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

PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters


def conv_algorithmic_flops(log, valid_by_axis):
    """Algorithmic FLOPs (valid tokens only, multiply-add = 2: SURVEY.md §8d) of the recorded conv-GEMM launches."""
    total = 0.0
    for kind, rows, n, cin, cout, taps in log:
        if kind != 'conv':
            continue
        valid = valid_by_axis.get(n, rows) if rows != n else rows
        total += 2.0 * taps * cin * cout * valid
    return total


def cpu_baseline(hp, n_threads):
    """The CPU oracle (port of the reference's op structure) on a bounded C1-shaped sample: 1 warm-up + 2 timed steps."""
    from oracle import daft_exprt_oracle as oracle
    from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict
    import ubisoft_laforge_daft_exprt_amd as pkg
    torch.set_num_threads(n_threads)
    shapes = {k: tuple(v.shape) for k, v in pkg.DaftExprt(hp.clone()).state_dict().items()}
    sd = synthetic_state_dict(shapes, 1234)
    for v in sd.values():
        v.requires_grad_(True)
    batch = synthetic_batch(n_speakers=hp.n_speakers, **CONFIGS['C1'])
    inputs = tuple(batch[i] for i in range(11)) + (batch[13],)
    targets = (batch[1], batch[3], batch[4], batch[8], batch[9], batch[10], batch[6], batch[7])
    frames = int(batch[9].sum())
    times = []
    for it in range(3):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = oracle.forward(sd, inputs, hp, training=True)
        total, _ = oracle.loss(out, targets, 1000, hp)
        total.backward()
        times.append(time.perf_counter() - t0)
        print(f'[cpu_baseline] step {it}: {times[-1]:.2f} s on {n_threads} threads', file=sys.stderr, flush=True)
        if sum(times) > 40.0 and it >= 1:                                 # keep the default run within minutes
            break
    timed = times[1:] if len(times) > 1 else times
    step = sum(timed) / len(timed)
    return {'value': frames / step, 'unit': 'mel frames/s', 'cores': n_threads, 'kind': 'port',
            'sample': f'C1 batch (B=4, {frames} valid frames), fwd+loss+bwd, dropout on, 1 warm-up + {len(timed)} timed steps, {step:.2f} s/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--precision', default='bf16', choices=['f32', 'bf16'])
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C5'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--no-dropout', action='store_true', help='diagnostic only: the headline metric is measured with dropout on')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import ubisoft_laforge_daft_exprt_amd as pkg
    from ubisoft_laforge_daft_exprt_amd import ops
    from ubisoft_laforge_daft_exprt_amd._lib import lib
    from ubisoft_laforge_daft_exprt_amd.ddp import GradientReducer
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
    from tests.helpers import manifest
    crit.load_pitch_predictor(synthetic_state_dict({k: tuple(v) for k, v in manifest()['pitch_predictor'].items()}, 1235))
    cfg = dict(CONFIGS[args.config])
    cfg['seed'] = cfg['seed'] + 1000 * rank                              # every rank owns different utterances
    cfg['n_speakers'] = n_speakers
    batch = synthetic_batch(**cfg)
    inputs, targets = model.parse_batch(dev, batch)
    targets = targets + (inputs[6], inputs[7])
    frames = int(batch[9].sum())
    symbols = int(batch[5].sum())
    reducer = GradientReducer(model, bucket_mb=16.0, grad_sink=os.environ.get('DX_NO_GRAD_SINK', '') == '')
    pkg.manual_seed(1234 + rank)

    def step(it):
        ops.invalidate_packs()                    # an optimiser step changes every weight ...
        ops.repack_all()                          # ... so every step re-packs them (one launch)
        reducer.zero_grad()
        out = model(inputs)
        total, _terms = crit(out, targets, it)
        total.backward()
        reducer.finish()
        return total

    for it in range(args.warmup):
        step(it)
    torch.cuda.synchronize()
    use_events = not args.no_kernel_events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        if use_events and it == args.steps - 1:
            # HIP-event bracketing of every conv-GEMM launch costs ~8 % of a step, so only the LAST timed step carries it
            lib().dx_prof_enable(0, 512)
            ops.record_launches(True)
        last = step(args.warmup + it)
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

    roofline = None
    if use_events:
        import ctypes
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        lib().dx_prof_collect(0, ctypes.cast(ctypes.pointer(n), ctypes.c_void_p), ctypes.cast(ctypes.pointer(ms), ctypes.c_void_p))
        log = ops.record_launches(False) or []
        flops = conv_algorithmic_flops(log, {int(batch[9].max()): frames, int(batch[5].max()): symbols})
        if n.value > 0 and ms.value > 0:
            achieved = flops / (ms.value * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(REPO, 'profiles', 'r01_conv_gemm_pmc.json')
            if os.path.exists(pmc):
                with open(pmc) as f:
                    traffic = json.load(f).get(args.precision, {}).get('hbm_bytes_per_launch')
            roofline = {'bound': 'mfma', 'kernel': 'conv-GEMM family: conv_ws_kernel / conv_dk_kernel / conv_gemm_kernel (Conv1d/Linear as MFMA GEMM, forward + input gradient)',
                        'achieved': round(achieved, 2), 'peak': PEAK_TFLOPS[args.precision], 'unit': 'TFLOP/s',
                        'frac': round(achieved / PEAK_TFLOPS[args.precision], 4), 'traffic': traffic,
                        'launches': n.value, 'avg_launch_us': round(1e3 * ms.value / n.value, 2),
                        'share_of_step': round(ms.value * 1e-3 / (elapsed / args.steps), 3),
                        'measured_on': 'every conv-GEMM launch of the last timed step'}

    if rank == 0:
        result = {
            'metric': 'mel frames/sec (fwd+bwd)', 'value': round(total_frames * args.steps / elapsed, 1), 'unit': 'valid mel frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if args.precision == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.config}: LJ-shaped training step, {cfg["batch_size"]} utterances/GPU, '
                                   f'L_max={int(batch[5].max())}, T_max={int(batch[9].max())}, {frames} valid frames/GPU/step; '
                                   'forward + loss (mel L1/L2, adversarial CE, post-mult, energy + pitch consistency) + backward'
                                   + (' + bucketed RCCL gradient all-reduce' if world > 1 else '') + (', dropout OFF (diagnostic)' if args.no_dropout else ', dropout on') + ', weights re-packed every step',
                       'operands': 'bf16 MFMA operands for Conv1d/Linear GEMMs and attention, fp32 accumulate, 1024-wide hidden tensors and qkv stored bf16'
                                   if args.precision == 'bf16' else 'exact f32 MFMA everywhere',
                       'parallelism': f'dp{world}'},
        }
        if roofline is not None:
            result['roofline'] = roofline
        if world == 1 and not args.no_cpu_baseline:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            cores = min(avail, 16)                                       # a 1-GPU box is given a 16-CPU share
            result['cpu_baseline'] = cpu_baseline(hp, cores)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
