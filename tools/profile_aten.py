"""Lists the ATen ops (glue around the HIP kernels) of one training step with the package call site that issued them."""
import collections, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import ubisoft_laforge_daft_exprt_amd as pkg
from ubisoft_laforge_daft_exprt_amd import ops
from ubisoft_laforge_daft_exprt_amd.ddp import GradientReducer
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch, synthetic_state_dict
from tests.helpers import manifest

dev = torch.device('cuda', 0)
pkg.set_precision('bf16')
hp = pkg.HyperParams(n_speakers=2)
model = pkg.DaftExprt(hp).to(dev)
model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234), strict=True)
model.train()
crit = pkg.DaftExprtLoss(dev, hp)
crit.load_pitch_predictor(synthetic_state_dict({k: tuple(v) for k, v in manifest()['pitch_predictor'].items()}, 1235))
cfg = dict(CONFIGS['C2']); cfg['n_speakers'] = 2
batch = synthetic_batch(**cfg)
inputs, targets = model.parse_batch(dev, batch)
targets = targets + (inputs[6], inputs[7])
reducer = GradientReducer(model, bucket_mb=16.0, grad_sink=True)
pkg.manual_seed(1234)

def step(it):
    model.runtime.invalidate_packs(); ops.repack_all(model.runtime); reducer.zero_grad()
    out = model(inputs); total, _ = crit(out, targets, it); total.backward(); reducer.finish()

for it in range(3): step(it)
torch.cuda.synchronize()
# python-level sources of zero fills / small elementwise glue: wrap the factories for one step
import traceback
srcs = collections.Counter()
def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'ubisoft_laforge_daft_exprt_amd' in fr.filename:
            return '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
    return 'other'
def _wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        srcs[(name, _site())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
    return orig
_o = [(_m, _n, _wrap(_m, _n)) for _m, _n in [(torch, 'zeros'), (torch, 'zeros_like'), (torch, 'empty'), (torch, 'empty_like'), (torch, 'where'), (torch, 'cat'), (torch, 'stack')]]
_oz = torch.Tensor.zero_
def _z(self): srcs[('zero_', _site())] += 1; return _oz(self)
torch.Tensor.zero_ = _z
_oc = torch.Tensor.contiguous
def _c(self, *a, **k):
    if not self.is_contiguous(): srcs[('contiguous(copy)', _site())] += 1
    return _oc(self, *a, **k)
torch.Tensor.contiguous = _c
_of = torch.Tensor.float
def _f(self, *a, **k):
    if self.dtype != torch.float32: srcs[('float(copy)', _site())] += 1
    return _of(self, *a, **k)
torch.Tensor.float = _f
step(3)
torch.cuda.synchronize()
for _m, _n, _orig in _o: setattr(_m, _n, _orig)
torch.Tensor.zero_ = _oz; torch.Tensor.contiguous = _oc; torch.Tensor.float = _of
print('python-level sources in one step:')
for k, v in sorted(srcs.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if k[0] not in ('empty', 'empty_like'): print('  %4d  %-18s %s' % (v, k[0], k[1]))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()
cnt = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.device_time_total > 0 and not any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        site = 'autograd/other'
        for fr in ev.stack:
            if 'ubisoft_laforge_daft_exprt_amd' in fr or 'bench.py' in fr or 'profile_aten' in fr:
                site = fr.split('ubisoft_laforge_daft_exprt_amd/')[-1]; break
        cnt[(ev.name, site)] += 1; dur[(ev.name, site)] += ev.device_time_total
tot = sum(dur.values())
print('aten device time per step: %.1f us over %d ops' % (tot, sum(cnt.values())))
for k, v in dur.most_common(45):
    print('%8.1f us %4d  %-28s %s' % (v, cnt[k], k[0], k[1]))
print('by op and input shapes:')
shp = collections.Counter(); shpd = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.device_time_total > 0 and not any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        k = (ev.name, str(ev.input_shapes)[:90])
        shp[k] += 1; shpd[k] += ev.device_time_total
for k, v in shpd.most_common(40):
    print('%8.1f us %4d  %-22s %s' % (v, shp[k], k[0], k[1]))
