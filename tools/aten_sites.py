"""Which ATen ops (the glue around the HIP path) one eager training step issues, by the package line that called them
(TorchDispatchMode; backward forced onto the calling thread so that the mode sees it)."""
import collections
import os
import sys
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                                                                  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode                                    # noqa: E402
import ubisoft_laforge_daft_exprt_amd as pkg                                                  # noqa: E402
from ubisoft_laforge_daft_exprt_amd.loss import pitch_predictor_shapes                        # noqa: E402
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict   # noqa: E402
from ubisoft_laforge_daft_exprt_amd.trainer import Trainer                                    # noqa: E402

VIEWS = ('view', 'as_strided', 'reshape', 'transpose', 'permute', 'slice', 'select', 'unsqueeze', 'squeeze', 'expand', 'detach', 'alias',
         't.default', 'unbind', 'split', '_unsafe_view', 'size', 'stride', 'numel', 'is_', 'empty', 'resize', 'set_', 'item', '_local_scalar',
         'lift_fresh', 'unflatten', 'flatten', 'narrow', 'chunk', 'storage_offset', 'dim', 'sym_', 'record_stream', 'is_pinned', '_to_copy_nop')


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = 'other'
            for fr in reversed(traceback.extract_stack()):
                if 'ubisoft_laforge_daft_exprt_amd/' in fr.filename:
                    site = f"{fr.filename.split('ubisoft_laforge_daft_exprt_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:90]}"
                    break
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            self.rows[(name, site, str(shapes))] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device('cuda', 0)
    pkg.set_precision('bf16')
    hp = pkg.HyperParams(n_speakers=2)
    model = pkg.DaftExprt(hp).to(dev)
    model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234), strict=True)
    crit = pkg.DaftExprtLoss(dev, hp)
    crit.load_pitch_predictor(synthetic_state_dict(pitch_predictor_shapes(), 1235))
    cfg = dict(CONFIGS['C2'])
    cfg['n_speakers'] = 2
    batch = synthetic_batch(**cfg)
    dev_batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in batch)
    for i in (5, 9):
        dev_batch[i]._dx_host_lengths = batch[i].tolist()
    trainer = Trainer(model, crit, hp, use_graphs=False)
    for _ in range(2):
        trainer.train_step([dev_batch])
    torch.cuda.synchronize()
    torch.autograd.set_multithreading_enabled(False)
    with Sites() as mode:
        trainer.train_step([dev_batch])
        torch.cuda.synchronize()
    for (name, site, shapes), n in sorted(mode.rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f'{n:3d} x {name:34s} {shapes:44s} {site}')
    print('total', sum(mode.rows.values()))
    return 0


if __name__ == '__main__':
    sys.exit(main())
