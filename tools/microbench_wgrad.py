"""The batched wide weight gradient (csrc/dx_gemm.hip: wgrad_bf16_kernel<3, true, true, 4>) alone, on the eight k = 3 jobs of the four
frame-level decoder blocks of the C2 workload (conv2: dY 128 wide, X = hidden 1024 wide; conv1: dY = hidden gradient, X 128 wide).
Use with DX_LIB_PATH=tools/ab/lib_<name>.so for ablation builds (tools/ablation_build.py <name> dx_gemm.hip -DDX_WG_ABL=n)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubisoft_laforge_daft_exprt_amd import ops
from ubisoft_laforge_daft_exprt_amd.synth import CONFIGS, synthetic_batch


def main():
    dev = 'cuda'
    ops.set_precision('bf16')
    rt = ops.DEFAULT
    rt.defer_wgrad = True
    batch = synthetic_batch(**CONFIGS['C2'])
    lens = batch[9].to(dev).to(torch.int32)
    B, N, D, Fc = lens.numel(), int(lens.max()), 128, 1024
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev)
    valid = (torch.arange(N, device=dev)[None, :] < lens[:, None]).float()[:, :, None]
    layers = []
    for _ in range(4):
        w1, w2 = rn(Fc, D, 3, sc=0.05), rn(D, Fc, 3, sc=0.02)
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        x = (rn(B, N, D) * valid).to(torch.bfloat16)
        dz = (rn(B, N, D) * valid).to(torch.bfloat16)
        h = (rn(B, N, Fc).relu() * valid).to(torch.bfloat16)
        dh = (rn(B, N, Fc) * valid).to(torch.bfloat16)
        sinks = [torch.zeros_like(w1), torch.zeros(Fc, device=dev), torch.zeros_like(w2), torch.zeros(D, device=dev)]
        layers.append((p1, p2, x, dz, h, dh, sinks))

    def launch():
        for p1, p2, x, dz, h, dh, s in layers:
            ops.conv_wgrad(dz, h, p2, lens, 0, w_sink=s[2], b_sink=s[3], defer=True)
            ops.conv_wgrad(dh, x, p1, lens, 1, w_sink=s[0], b_sink=s[1], defer=True)
        return ops.flush_wgrads(rt)

    n_launch = launch()
    torch.cuda.synchronize()
    if os.environ.get('DX_LIB_PATH') is None:      # one check against torch (first layer, conv1), the default build only
        p1, p2, x, dz, h, dh, s = layers[0]
        xf, dhf = x.float(), dh.float()
        xp = torch.nn.functional.pad(xf, (0, 0, 1, 1))
        ref = torch.stack([torch.einsum('bnc,bnd->cd', dhf, xp[:, t:t + N]) for t in range(3)], dim=-1)
        err = ((s[0] - ref).abs().max() / ref.abs().max()).item()
        print(f'conv1 dW vs torch: max rel {err:.2e}')
    # graph replay: the eight queue calls + the flush cost ~100 us of host time, more than some ablation builds run
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    n = 20
    a.record()
    for _ in range(n):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    tokens = int(lens.sum())
    flops = 8 * 2 * 3 * D * Fc * tokens
    byts = 4 * 2 * tokens * (Fc + D) * 2
    print(f'{os.environ.get("DX_LIB_PATH", "default"):40s} {n_launch} launch: {us:7.1f} us   {flops / us / 1e6:6.1f} TF/s   {byts / us / 1e3:6.1f} GB/s algorithmic')


if __name__ == '__main__':
    main()
