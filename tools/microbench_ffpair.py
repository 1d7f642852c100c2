"""Fused conv feed-forward pair (csrc/dx_ffpair.hip) vs the two launches it replaces, at the C2 frame-level shape."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubisoft_laforge_daft_exprt_amd import ops
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch


def timeit(fn, n=30):
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
    lens = (batch[9] if axis == 'frame' else batch[5]).to(dev).to(torch.int32)
    B, N, D, Fc = lens.numel(), int(lens.max()), 128, 1024
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev)
    w1, b1 = rn(Fc, D, 3, sc=1 / math.sqrt(3 * D)), rn(Fc, sc=0.1)
    w2, b2 = rn(D, Fc, 3, sc=1 / math.sqrt(3 * Fc)), rn(D, sc=0.1)
    p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
    valid = (torch.arange(N, device=dev)[None, :] < lens[:, None]).float()[:, :, None]
    x = (rn(B, N, D) * valid).to(torch.bfloat16)
    dz = (rn(B, N, D) * valid).to(torch.bfloat16)
    base = rn(B, N, D)
    h = ops.conv_gemm(x, p1, b1, relu=True, lens=lens, halo=1, out_dtype=torch.bfloat16)
    tokens = int(lens.sum())
    flops = 2 * 2 * 3 * D * Fc * tokens

    def unfused_fwd():
        hh = ops.conv_gemm(x, p1, b1, relu=True, lens=lens, halo=1, out_dtype=torch.bfloat16)
        return ops.conv_gemm(hh, p2, b2, lens=lens, halo=0)

    def unfused_bwd():
        dh = ops.conv_gemm(dz, p2, None, transpose=True, relu_aux=h, lens=lens, halo=1, out_dtype=torch.bfloat16)
        return ops.conv_gemm(dh, p1, None, transpose=True, out=base, accumulate=True, lens=lens, halo=0)

    rows = [('two launches fwd', timeit(unfused_fwd)), ('fused fwd', timeit(lambda: ops.ff_pair(x, p1, p2, b1, b2, lens))),
            ('two launches bwd', timeit(unfused_bwd)),
            ('fused bwd', timeit(lambda: ops.ff_pair(dz, p1, p2, None, None, lens, backward=True, aux=h, out=base, accumulate=True)))]
    # the forms the model launches: + LayerNorm epilogue (+ the next block's q/k/v), and the whole feed-forward half of the block's backward
    ln_w, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    film = rn(B, 2 * D)
    win, bin_ = rn(3 * D, D, sc=1 / math.sqrt(D)), rn(3 * D, sc=0.1)
    pin = ops.PackedWeight(win)
    wout = rn(D, D, sc=1 / math.sqrt(D))
    pout = ops.PackedWeight(wout)
    res = rn(B, N, D)
    rows.append(('fwd + LN epilogue', timeit(lambda: ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, ln_w, ln_b, film, seed_pre=5, p_pre=0.1))))
    rows.append(('fwd + LN + next qkv', timeit(lambda: ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, ln_w, ln_b, film, seed_pre=5, p_pre=0.1, next_in=(pin, bin_)))))
    z2, _, _, mean2, rstd2, hmask = ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, ln_w, ln_b, film, seed_pre=5, p_pre=0.1, want_mask=True)
    rows.append(('fwd + LN + mask out', timeit(lambda: ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, ln_w, ln_b, film, seed_pre=5, p_pre=0.1, want_mask=True))))
    z1, mean1, rstd1 = rn(B, N, D), rn(B, N), rn(B, N).abs() + 0.5
    dy2 = rn(B, N, D) * valid
    rows.append(('block bwd (LN2b+pair+LN1b)', timeit(lambda: ops.ff_block_bwd(dy2, z2, mean2, rstd2, ln_w, ln_b, film, p1, p2, lens, h, z1, mean1, rstd1, ln_w, ln_b,
                                                                               seed2=5, p2=0.1, seed1=6, p1=0.1))))
    rows.append(('block bwd + datt', timeit(lambda: ops.ff_block_bwd(dy2, z2, mean2, rstd2, ln_w, ln_b, film, p1, p2, lens, h, z1, mean1, rstd1, ln_w, ln_b,
                                                                     seed2=5, p2=0.1, seed1=6, p1=0.1, out_pack=pout))))
    rows.append(('block bwd + datt, sign words', timeit(lambda: ops.ff_block_bwd(dy2, z2, mean2, rstd2, ln_w, ln_b, film, p1, p2, lens, h, z1, mean1, rstd1, ln_w, ln_b,
                                                                                 seed2=5, p2=0.1, seed1=6, p1=0.1, out_pack=pout, hmask=hmask))))
    for name, us in rows:
        print(f'{axis}-level B={B} N={N} valid tokens={tokens}: {name:28s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s algorithmic '
              f'({100 * flops / us / 1e6 / 2500:.1f} % of 2.5 PF)')


if __name__ == '__main__':
    main()
