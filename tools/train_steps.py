"""Training-step harness: feature files (or synthetic batches) -> on-device conditioning -> forward / loss / backward with
micro-batch accumulation -> bucketed gradient exchange -> fused Adam with clipping and the LR schedule (Trainer.train_step,
the structure of the reference's src/daft_exprt/train.py:380-539).

    python tools/train_steps.py --steps 20 --precision bf16 [--accumulation 2] [--features LIST_FILE --batch-size 16]
    python -m torch.distributed.run --nproc-per-node N tools/train_steps.py ...        (one rank per GPU, RCCL)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--precision', default='bf16', choices=['f32', 'bf16'])
    ap.add_argument('--accumulation', type=int, default=1)
    ap.add_argument('--batch-size', type=int, default=16)
    ap.add_argument('--features', default='', help='list file in the reference format (features_dir|feature_file|speaker_id per line)')
    ap.add_argument('--no-dropout', action='store_true')
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    import ubisoft_laforge_daft_exprt_amd as pkg
    from ubisoft_laforge_daft_exprt_amd.loss import pitch_predictor_shapes
    from ubisoft_laforge_daft_exprt_amd.synth import synthetic_batch, synthetic_state_dict
    from ubisoft_laforge_daft_exprt_amd.trainer import Trainer
    pkg.set_precision(args.precision)
    hp = pkg.HyperParams(n_speakers=3, accumulation_steps=args.accumulation)
    if args.no_dropout:
        hp = hp.without_dropout()
    model = pkg.DaftExprt(hp).to(dev)
    model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234 + rank))   # broadcast fixes it
    crit = pkg.DaftExprtLoss(dev, hp)
    crit.load_pitch_predictor(synthetic_state_dict(pitch_predictor_shapes(), 1235))
    trainer = Trainer(model, crit, hp)
    pkg.manual_seed(1234 + rank)
    if args.features:
        from ubisoft_laforge_daft_exprt_amd.features import FeatureSet
        data = iter(FeatureSet(args.features, hp, args.batch_size, rank=rank, world=world))
        next_batch = lambda i: next(data)
    else:
        next_batch = lambda i: synthetic_batch(args.batch_size, (20, 60), seed=100 + 7 * i + 1000 * rank, n_speakers=3)
    t0 = time.perf_counter()
    for step in range(args.steps):
        micro = [next_batch(step * args.accumulation + k) for k in range(args.accumulation)]
        lr = trainer.learning_rate
        loss, terms, norm = trainer.train_step(micro)
        if rank == 0:
            value = float(loss)
            trainer.note_loss(value)
            print(json.dumps({'iteration': trainer.iteration - 1, 'loss': round(value, 6), 'grad_norm': round(float(norm), 5), 'lr': lr,
                              'mel_l1': round(sum(t['mel_spec_l1_loss'] for t in terms) / len(terms), 6)}), flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        print(f'{args.steps} steps in {time.perf_counter() - t0:.2f} s (incl. host batch synthesis and per-step loss fetch)', file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
