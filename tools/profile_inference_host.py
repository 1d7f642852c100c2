"""cProfile of the host side of one C4 inference call (diagnostic)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ubisoft_laforge_daft_exprt_amd as pkg
from ubisoft_laforge_daft_exprt_amd.inference import GraphedSynthesizer
from ubisoft_laforge_daft_exprt_amd.synth import synthetic_inference_batch, synthetic_state_dict

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
pkg.set_precision(prec)
hp = pkg.HyperParams(n_speakers=2, stats={'spk 0': {'pitch': {'mean': 5.0, 'std': 0.25}}})
model = pkg.DaftExprt(hp).to('cuda')
model.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1234))
synth = GraphedSynthesizer(model, hp)
inputs, prosody, spk, accent = synthetic_inference_batch()
mv = lambda t: t.clone().to('cuda')
args = lambda: (tuple(mv(t) for t in inputs), 'add', {k: mv(v) for k, v in prosody.items()}, mv(spk), mv(accent))
for mode in (True, False):
    for _ in range(3):
        synth(*args(), use_graph=mode)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        synth(*args(), use_graph=mode)
    torch.cuda.synchronize()
    pr.disable()
    print('==== use_graph =', mode)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
