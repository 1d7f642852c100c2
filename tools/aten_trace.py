"""Which ATen kernels (glue around the HIP path) run in one training step, by the package line that issued them (torch.profiler
with stacks, forward and autograd threads)."""
import collections, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from torch.profiler import profile, ProfilerActivity
import ubisoft_laforge_daft_exprt_amd as pkg
from ubisoft_laforge_daft_exprt_amd.loss import pitch_predictor_shapes
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict
from ubisoft_laforge_daft_exprt_amd.trainer import Trainer

dev = torch.device('cuda', 0)
pkg.set_precision('bf16')
hp = pkg.HyperParams(n_speakers=2)
model = pkg.DaftExprt(hp).to(dev)
model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234), strict=True)
crit = pkg.DaftExprtLoss(dev, hp)
crit.load_pitch_predictor(synthetic_state_dict(pitch_predictor_shapes(), 1235))
cfg = dict(CONFIGS['C2']); cfg['n_speakers'] = 2
batch = synthetic_batch(**cfg)
dev_batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in batch)
for i in (5, 9):
    dev_batch[i]._dx_host_lengths = batch[i].tolist()
trainer = Trainer(model, crit, hp, use_graphs=False)
for _ in range(3):
    trainer.train_step([dev_batch])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_step([dev_batch])
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_device_time_total', row_limit=45, max_name_column_width=40, max_src_column_width=110))
rows = collections.Counter()
times = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or ev.cpu_children and any(c.device_time_total > 0 for c in ev.cpu_children):
        continue
    site = 'autograd / other'
    for fr in ev.stack:
        if 'ubisoft_laforge_daft_exprt_amd' in fr:
            site = fr.split('ubisoft_laforge_daft_exprt_amd/')[-1]
            break
    rows[(ev.name, site)] += 1
    times[(ev.name, site)] += ev.device_time_total
tot = 0
for (name, site), n in sorted(rows.items(), key=lambda kv: -times[kv[0]]):
    print(f'{n:4d} x {name:28s} {times[(name, site)]:8.1f} us   {site}')
    tot += times[(name, site)]
print(f'total device time of leaf ATen ops: {tot:.1f} us in {sum(rows.values())} launches')
