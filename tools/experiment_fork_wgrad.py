"""ONE designed run of the round-2 anomaly ("something unexplained"): the frame decoder's queued k = 1 weight
gradients launched on a SIDE stream where the backward drops to the symbol level, beside the upsampler's backward on the compute stream.
Three arms on the same C2 batch, eager, dropout off, 6 steps each, every gradient compared with the serial step's:
  fork        the experiment as it was run in round 2 (the queue's tensor references are dropped right after the side-stream launch)
  fork+keep   the same, but the tensors the side-stream kernels READ stay referenced until the streams are joined
  fork+record the same as 'fork', but every operand is record_stream()-ed on the side stream
If 'fork' differs and the other two do not, the cause is the caching allocator handing the operands' memory (freed in compute-stream
order) to the compute stream's next kernels while the side-stream kernels still read it -- a host-side lifetime bug of the experiment,
not a device-side lost atomic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ubisoft_laforge_daft_exprt_amd as pkg
from ubisoft_laforge_daft_exprt_amd import ops
from ubisoft_laforge_daft_exprt_amd.loss import pitch_predictor_shapes
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict
from ubisoft_laforge_daft_exprt_amd.trainer import Trainer

dev = 'cuda'
pkg.set_precision('bf16')
hp = pkg.HyperParams(n_speakers=2).without_dropout()
model = pkg.DaftExprt(hp).to(dev)
model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234), strict=True)
crit = pkg.DaftExprtLoss(dev, hp)
crit.load_pitch_predictor(synthetic_state_dict(pitch_predictor_shapes(), 1235))
t = Trainer(model, crit, hp, use_graphs=False, cuts=0)
batch = synthetic_batch(**CONFIGS['C2'])
dev_batch = tuple(x.to(dev) if torch.is_tensor(x) else x for x in batch)
parsed, _ = t._parse([dev_batch])
rt = model.runtime
side = torch.cuda.Stream()
mode = {'arm': 'serial', 'keep': []}
orig_upsample_bwd = ops.upsample_bwd


def upsample_bwd(*a, **kw):
    """first call of GaussianUpsampleFn.backward: the decoder's backward has been issued -- fork its k = 1 weight gradients here"""
    if mode['arm'] != 'serial':
        q = rt.wgrad_queue
        k1 = {k: v for k, v in q.items() if k[0] == 1}
        for k in k1:
            del q[k]
        if mode['arm'] == 'fork+keep':
            mode['keep'].append([ten for jobs in k1.values() for _, ten in jobs])
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if mode['arm'] == 'fork+record':
                for jobs in k1.values():
                    for _, tens in jobs:
                        for x in tens:
                            if torch.is_tensor(x):
                                x.record_stream(side)
            rest, rt.wgrad_queue = rt.wgrad_queue, k1
            ops.flush_wgrads(rt)
            rt.wgrad_queue = rest
    return orig_upsample_bwd(*a, **kw)


ops.upsample_bwd = upsample_bwd


def step(arm):
    mode['arm'] = arm
    t._phases(parsed, 4000, t.reducer.launch_group)
    torch.cuda.current_stream().wait_stream(side)
    mode['keep'].clear()
    t.reducer.finish()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}


serial = step('serial')
again = step('serial')
noise = max(float((again[k] - serial[k]).abs().max() / serial[k].abs().max().clamp_min(1e-30)) for k in serial)
print(f'serial vs serial (atomics order only): worst relative difference {noise:.2e}', flush=True)
for arm in ('fork', 'fork+keep', 'fork+record', 'fork'):
    for rep in range(6):
        got = step(arm)
        rows = sorted(((float((got[k] - serial[k]).abs().max() / serial[k].abs().max().clamp_min(1e-30)), k) for k in serial), reverse=True)
        bad = [(f'{r:.1e}', k) for r, k in rows if r > 2e-3]
        line = f'{arm:12s} step {rep}: worst {rows[0][0]:.2e} at {rows[0][1]}; parameters off by > 2e-3: {len(bad)}'
        if bad:
            k = rows[0][1]
            d = ((got[k] - serial[k]).abs() > 1e-3 * serial[k].abs().max()).flatten().nonzero().flatten().tolist()
            line += f'; {bad[:6]}; differing elements of {k} (first 24 of {len(d)}): {d[:24]}'
        print(line, flush=True)
