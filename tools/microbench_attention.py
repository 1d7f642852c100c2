"""The three bf16 attention kernels at the C2 frame-level shape (B = 48, N = 896, 2 heads x 64), 16-bit q/k/v, context and gradients:
microseconds per launch with and without dropout, with the C2 lengths and with every utterance at full length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubisoft_laforge_daft_exprt_amd import ops
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = 'cuda'
    ops.set_precision('bf16')
    axis = sys.argv[1] if len(sys.argv) > 1 else 'frame'
    batch = synthetic_batch(**CONFIGS['C2'])
    lens_c2 = (batch[9] if axis == 'frame' else batch[5]).to(dev).to(torch.int32)
    B, N = lens_c2.numel(), int(lens_c2.max())
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, N, 384, generator=g).to(dev).to(torch.bfloat16)
    dctx = torch.randn(B, N, 128, generator=g).to(dev).to(torch.bfloat16)
    h = torch.bfloat16
    for tag, lens in (('C2 lengths', lens_c2), ('all full', torch.full_like(lens_c2, N))):
        tiles = sum(((int(l) + 63) // 64) ** 2 for l in lens.tolist()) * 2
        for p in (0.1, 0.0):
            ctx, lse = ops.attention_fwd(qkv, lens, 2, 11, p, ctx_dtype=h)
            t_f = timeit(lambda: ops.attention_fwd(qkv, lens, 2, 11, p, ctx_dtype=h))
            t_b = timeit(lambda: ops.attention_bwd(qkv, ctx, dctx, lse, lens, 2, 11, p, out_dtype=h))
            print(f'{axis} {tag:10s} p={p}: fwd {t_f:6.1f} us  bwd(dq+dkv) {t_b:6.1f} us   ({tiles} 64x64 tiles: fwd {1e3 * t_f / tiles * 256:.0f} ns/tile/CU-slot)', flush=True)


if __name__ == '__main__':
    main()
