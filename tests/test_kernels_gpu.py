"""Kernel-level parity: every C-ABI entry point against plain PyTorch fp32 math on the same inputs (GPU box only)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    from ubisoft_laforge_daft_exprt_amd import ops as _ops
    _ops.set_precision('f32')
    return _ops


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).to(DEV)


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def lens_tensor(vals):
    return torch.tensor(vals, dtype=torch.int32, device=DEV)


def ref_conv(x, w, b, taps):
    if taps == 1:
        return F.linear(x, w, b)
    return F.conv1d(x.transpose(1, 2), w, b, padding=1).transpose(1, 2)


@pytest.mark.parametrize('precision,tol', [('f32', 5e-6), ('bf16', 2e-2), ('fp16', 3e-3)])
@pytest.mark.parametrize('B,N,Cin,Cout,taps', [(3, 150, 80, 1024, 3), (2, 129, 1024, 128, 3), (4, 37, 128, 384, 1),
                                                (5, 1, 192, 128, 1), (2, 300, 128, 80, 1), (3, 17, 128, 3, 1)])
def test_conv_gemm_forward_dgrad_wgrad(ops, precision, tol, B, N, Cin, Cout, taps):
    ops.set_precision(precision)
    try:
        wshape = (Cout, Cin, 3) if taps == 3 else (Cout, Cin)
        w = randn(*wshape, seed=1, scale=1.0 / math.sqrt(Cin * taps)).requires_grad_(True)
        b = randn(Cout, seed=2, scale=0.1).requires_grad_(True)
        x = randn(B, N, Cin, seed=3).requires_grad_(True)
        pack = ops.PackedWeight(w)
        y = ops.conv_gemm(x.detach(), pack, b.detach())
        y_ref = ref_conv(x, w, b, taps)
        assert rel_err(y, y_ref) < tol
        dy = randn(B, N, Cout, seed=4)
        y_ref.backward(dy)
        if Cout % 4 == 0:
            dx = ops.conv_gemm(dy, pack, None, transpose=True)
            assert rel_err(dx, x.grad) < tol
            if precision == 'f32' or (Cin % 8 == 0 and Cout % 8 == 0):
                dw, db = ops.conv_wgrad(dy, x.detach(), pack)
                assert rel_err(dw, w.grad) < {'f32': 1e-5, 'bf16': 2e-2, 'fp16': 3e-3}[precision]
                assert rel_err(db, b.grad) < (1e-5 if precision == 'f32' else 1e-2)     # fused bias gradient
                assert rel_err(ops.colsum(dy), b.grad) < 1e-5
    finally:
        ops.set_precision('f32')


def test_batched_weight_gradients_many_layers_in_one_launch(ops):
    """More layers than one round of workgroups holds (dx_conv_wgrad_batched takes 32): 20 k = 1 layers of three shapes, one launch,
    each against fp32 torch on the rounded operands."""
    ops.set_precision('bf16')
    try:
        rt = ops.DEFAULT
        rt.defer_wgrad = True
        jobs = []
        for i in range(20):
            B, N, Cin, Cout = [(2, 100, 128, 128), (3, 77, 128, 384), (2, 130, 256, 128)][i % 3]
            w = randn(Cout, Cin, seed=100 + i, scale=0.05)
            pack = ops.PackedWeight(w)
            x, dy = randn(B, N, Cin, seed=200 + i).to(torch.bfloat16), randn(B, N, Cout, seed=300 + i).to(torch.bfloat16)
            gw, gb = torch.zeros_like(w), torch.zeros(Cout, device=DEV)
            assert ops.conv_wgrad(dy, x, pack, None, -1, w_sink=gw, b_sink=gb, defer=True) == (None, None)
            jobs.append((dy, x, gw, gb))
        assert ops.flush_wgrads(rt) == 1
        for dy, x, gw, gb in jobs:
            ref = torch.einsum('bnc,bnd->cd', dy.float(), x.float())
            assert rel_err(gw, ref) < 1e-4 and rel_err(gb, dy.float().sum((0, 1))) < 1e-4
    finally:
        rt.defer_wgrad = False
        rt.wgrad_queue.clear()
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('taps', [1, 3])
def test_batched_weight_gradients_equal_single_launches(ops, precision, taps):
    """dx_conv_wgrad_batched: queued layers of different shapes / lengths / halo rules in one launch == one dx_conv_wgrad launch per
    layer (same kernel, other token split: equal up to fp32 summation order) and == the torch reference within the mode's tolerance."""
    ops.set_precision(precision)
    try:
        h16 = {'bf16': torch.bfloat16, 'fp16': torch.float16}[precision]
        rt = ops.DEFAULT
        geo = [(3, 200, 128, 1024, [200, 131, 64], 1), (3, 200, 1024, 128, [200, 131, 64], 0), (2, 333, 128, 384, [333, 17], 0),
               (3, 200, 128, 128, [200, 131, 64], -1), (2, 90, 256, 136, [90, 90], 0)]
        jobs = []
        for i, (B, N, Cin, Cout, lens, halo) in enumerate(geo):
            w = randn(*((Cout, Cin, 3) if taps == 3 else (Cout, Cin)), seed=10 + i, scale=0.05)
            pack = ops.PackedWeight(w)
            x = randn(B, N, Cin, seed=20 + i).to(h16)
            dy = randn(B, N, Cout, seed=30 + i).to(h16)
            L = lens_tensor(lens)
            if halo >= 0:                                  # the contract of skip_halo: rows beyond len + halo are zero (whole chunks of them are skipped)
                for b, n in enumerate(lens):
                    dy[b, n + halo:] = 0
            ref_w, ref_b = ops.conv_wgrad(dy, x, pack, L if halo >= 0 else None, halo)
            jobs.append((dy, x, pack, L if halo >= 0 else None, halo, ref_w, ref_b))
        rt.defer_wgrad = True
        sinks = []
        for dy, x, pack, L, halo, _, _ in jobs:
            gw, gb = torch.zeros_like(pack.weight), torch.zeros(pack.cout, device=DEV)
            assert ops.conv_wgrad(dy, x, pack, L, halo, w_sink=gw, b_sink=gb, defer=True) == (None, None)
            sinks.append((gw, gb))
        assert all(float(gw.abs().max()) == 0.0 for gw, _ in sinks)                     # nothing launched yet
        assert ops.flush_wgrads(rt) == 1 and not rt.wgrad_queue
        for (dy, x, pack, L, halo, ref_w, ref_b), (gw, gb) in zip(jobs, sinks):
            assert rel_err(gw, ref_w) < 2e-5 and rel_err(gb, ref_b) < 2e-5, (pack.cin, pack.cout)
        # and against fp32 torch on the same 16-bit-rounded operands: every job (ragged ends inside a 64-token chunk, the halo row of
        # the first / last chunk of an utterance, a partial output-channel tile, two input-channel tiles)
        for (dy, x, pack, L, halo, ref_w, ref_b), (gw, gb) in zip(jobs, sinks):
            wr = pack.weight.detach().clone().requires_grad_(True)
            br = torch.zeros(pack.cout, device=DEV, requires_grad=True)
            ref_conv(x.float(), wr, br, taps).backward(dy.float())
            assert rel_err(gw, wr.grad) < 1e-4 and rel_err(gb, br.grad) < 1e-4, (pack.cin, pack.cout)
    finally:
        ops.DEFAULT.defer_wgrad = False
        ops.DEFAULT.wgrad_queue.clear()
        ops.set_precision('f32')


def test_conv_gemm_epilogues(ops):
    B, N, Cin, Cout = 2, 70, 128, 256
    w = randn(Cout, Cin, 3, seed=1, scale=0.05)
    b = randn(Cout, seed=2, scale=0.1)
    big = randn(B, N, Cin + 64, seed=3)
    x = big[:, :, 32:32 + Cin]                       # strided view: ldx > Cin
    lens = lens_tensor([70, 41])
    sc, sh = randn(Cout, seed=5), randn(Cout, seed=6)
    aux = randn(B, N, Cout, seed=7)
    pack = ops.PackedWeight(w)
    y = ops.conv_gemm(x, pack, b, relu=True, post_scale=sc, post_shift=sh, relu_aux=aux, lens=lens, mask_rows=True, out_scale=-0.5)
    ref = F.relu(ref_conv(x, w, b, 3)) * sc + sh
    ref = ref * (aux > 0) * -0.5
    ref[1, 41:] = 0
    assert rel_err(y, ref) < 2e-6
    y2 = ops.conv_gemm(x, pack, b, out=y.clone(), accumulate=True)
    assert rel_err(y2, y + ref_conv(x, w, b, 3)) < 2e-6
    with pytest.raises(RuntimeError):
        ops.conv_gemm(randn(2, 5, 126), ops.PackedWeight(randn(8, 126)), None)   # Cin not a multiple of 4


def test_bf16_hidden_storage(ops):
    """bf16 operand mode stores the 1024-wide hidden activations as bf16: conv -> bf16 out, bf16 in -> conv, bf16 relu mask, bf16 wgrad."""
    ops.set_precision('bf16')
    try:
        B, N, D, Fc = 3, 200, 128, 1024
        lens = lens_tensor([200, 131, 64])
        w1 = randn(Fc, D, 3, seed=1, scale=0.05).requires_grad_(True)
        w2 = randn(D, Fc, 3, seed=2, scale=0.02).requires_grad_(True)
        b1 = randn(Fc, seed=3, scale=0.1)
        x = randn(B, N, D, seed=4).requires_grad_(True)
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        h = ops.conv_gemm(x.detach(), p1, b1, relu=True, out_dtype=torch.bfloat16)
        assert h.dtype == torch.bfloat16
        z = ops.conv_gemm(h, p2, None)
        h_ref = F.relu(ref_conv(x, w1, b1, 3))
        z_ref = ref_conv(h_ref, w2, None, 3)
        assert rel_err(h.float(), h_ref) < 2e-2 and rel_err(z, z_ref) < 2e-2
        dz = randn(B, N, D, seed=5)
        # reference backward THROUGH THE KERNEL'S OWN ReLU MASK (a bf16 forward flips the sign of near-zero pre-activations,
        # which is inherent to bf16 compute and would otherwise dominate the comparison)
        hk = h.float().requires_grad_(True)
        ref_conv(hk, w2, None, 3).backward(dz)
        dh_ref = hk.grad * (h > 0)
        dw2_ref = torch.autograd.grad(ref_conv(h.float(), w2, None, 3), w2, dz)[0]
        ref_conv(x, w1, b1, 3).backward(dh_ref)
        dh = ops.conv_gemm(dz, p2, None, transpose=True, relu_aux=h, out_dtype=torch.bfloat16)
        assert rel_err(dh.float(), dh_ref) < 2e-2
        dx = ops.conv_gemm(dh, p1, None, transpose=True)
        assert rel_err(dx, x.grad) < 3e-2
        assert rel_err(ops.conv_wgrad(dz, h, p2)[0], dw2_ref) < 3e-2
        assert rel_err(ops.conv_wgrad(dh, x.detach(), p1)[0], w1.grad) < 3e-2
        assert rel_err(ops.colsum(dh), (dh.float()).sum((0, 1))) < 1e-4
        # tile skipping keeps skipped tiles defined (zeros) for bf16 outputs too
        h2 = ops.conv_gemm(x.detach(), p1, b1, relu=True, out_dtype=torch.bfloat16, lens=lens, halo=1)
        assert torch.equal(h2[2, 128:], torch.zeros_like(h2[2, 128:])) and torch.equal(h2[0], h[0]) and torch.equal(h2[2, :128], h[2, :128])
        dzm = dz * (torch.arange(N, device=DEV)[None, :, None] < lens[:, None, None])
        assert rel_err(ops.conv_wgrad(dzm, h, p2, lens, 0)[0], ops.conv_wgrad(dzm, h, p2)[0]) < 1e-5
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('Cin,Cout,taps', [(128, 1024, 3), (128, 384, 1), (128, 128, 1), (80, 1024, 3), (128, 80, 1)])
def test_weight_stationary_conv_path(ops, Cin, Cout, taps):
    """bf16 layers with Cin <= 128 and >= 64 token tiles run the persistent weight-stationary kernel: check it against the tiled
    kernel (DX_CONV_WS is read once per process, so the reference here is PyTorch) for every epilogue variant."""
    ops.set_precision('bf16')
    try:
        B, N = 9, 1000                                       # 9 x 8 = 72 token tiles, ragged last tile
        lens = lens_tensor([1000, 999, 897, 896, 640, 513, 129, 128, 1])
        wshape = (Cout, Cin, 3) if taps == 3 else (Cout, Cin)
        w = randn(*wshape, seed=1, scale=1.0 / math.sqrt(Cin * taps))
        b = randn(Cout, seed=2, scale=0.1)
        x = randn(B, N, Cin, seed=3)
        pack = ops.PackedWeight(w)
        ref = ref_conv(x, w, b, taps)
        assert rel_err(ops.conv_gemm(x, pack, b), ref) < 2e-2
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
        y = ops.conv_gemm(x, pack, b, lens=lens, mask_rows=True, halo=0)
        assert rel_err(y, ref * valid) < 2e-2 and torch.equal(y * ~valid, torch.zeros_like(y))
        if Cout % 8 == 0:
            aux = randn(B, N, Cout, seed=4).to(torch.bfloat16)
            sc, sh = 1 + 0.1 * randn(Cout, seed=5), torch.zeros(Cout, device=DEV)
            yb = ops.conv_gemm(x, pack, b, relu=True, post_scale=sc, post_shift=sh, relu_aux=aux, lens=lens, halo=1, out_dtype=torch.bfloat16,
                               out_scale=0.5)
            refb = (F.relu(ref) * sc) * 0.5 * (aux.float() > 0)
            keep = (torch.arange(N, device=DEV)[None, :, None] < ((lens[:, None, None] + 1 + 127) // 128) * 128)   # whole tiles inside len+halo
            assert yb.dtype == torch.bfloat16 and rel_err(yb.float() * keep, refb * keep) < 2e-2
            assert torch.equal(yb.float() * ~keep, torch.zeros_like(refb))
        acc0 = randn(B, N, Cout, seed=6)
        ya = ops.conv_gemm(x, pack, None, out=acc0.clone(), accumulate=True, lens=lens, halo=0)
        tiles_ok = (torch.arange(N, device=DEV)[None, :, None] < ((lens[:, None, None] + 127) // 128) * 128)
        assert rel_err(ya * tiles_ok, (acc0 + ref_conv(x, w, None, taps)) * tiles_ok) < 2e-2
        assert torch.equal(ya * ~tiles_ok, acc0 * ~tiles_ok)               # skipped tiles are left untouched when accumulating
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('Cin,Cout,taps,x_bf16', [(1024, 128, 3, True), (1024, 128, 3, False), (320, 256, 3, True), (256, 384, 1, False),
                                                   (1024, 80, 1, True), (448, 1024, 3, True)])
def test_deep_k_conv_path(ops, Cin, Cout, taps, x_bf16):
    """bf16 layers with Cin >= 256 run the staggered two-waves-per-SIMD kernel (K split inside the workgroup, LDS double buffer):
    odd and even stage counts, both activation storage types, every epilogue variant, tile skipping."""
    ops.set_precision('bf16')
    try:
        B, N = 4, 300                                        # 3 token tiles per row, ragged last tile
        lens = lens_tensor([300, 257, 128, 5])
        wshape = (Cout, Cin, 3) if taps == 3 else (Cout, Cin)
        w = randn(*wshape, seed=1, scale=1.0 / math.sqrt(Cin * taps))
        b = randn(Cout, seed=2, scale=0.1)
        x = randn(B, N, Cin, seed=3)
        xin = x.to(torch.bfloat16) if x_bf16 else x
        xr = xin.float()
        pack = ops.PackedWeight(w)
        ref = ref_conv(xr, w, b, taps)
        assert rel_err(ops.conv_gemm(xin, pack, b), ref) < 1e-2
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
        y = ops.conv_gemm(xin, pack, b, lens=lens, mask_rows=True, halo=0)
        assert rel_err(y, ref * valid) < 1e-2 and torch.equal(y * ~valid, torch.zeros_like(y))
        if Cout % 8 == 0:
            aux = randn(B, N, Cout, seed=4).to(torch.bfloat16)
            sc, sh = 1 + 0.1 * randn(Cout, seed=5), 0.1 * randn(Cout, seed=7)
            yb = ops.conv_gemm(xin, pack, b, relu=True, post_scale=sc, post_shift=sh, relu_aux=aux, lens=lens, halo=1, out_dtype=torch.bfloat16,
                               out_scale=0.5)
            refb = (F.relu(ref) * sc + sh) * 0.5 * (aux.float() > 0)
            # the contract of skip_halo: rows below len + halo are computed; whole tiles beyond them are zero-filled.  The tile is the
            # kernel's own business (128 tokens, 64 in the short-token kernel): rows between len + halo and the next 128 boundary may be either
            n_ax = torch.arange(N, device=DEV)[None, :, None]
            must, beyond = n_ax < lens[:, None, None] + 1, n_ax >= ((lens[:, None, None] + 1 + 127) // 128) * 128
            assert yb.dtype == torch.bfloat16 and rel_err(yb.float() * must, refb * must) < 2e-2
            assert torch.equal(yb.float() * beyond, torch.zeros_like(refb))
            between = ~must & ~beyond
            assert bool(((yb.float() == 0) | ((yb.float() - refb).abs() <= 2e-2 * refb.abs().max()))[between.expand_as(refb)].all())
        acc0 = randn(B, N, Cout, seed=6)
        ya = ops.conv_gemm(xin, pack, None, out=acc0.clone(), accumulate=True, lens=lens, halo=0)
        n_ax = torch.arange(N, device=DEV)[None, :, None]
        must, beyond = n_ax < lens[:, None, None], n_ax >= ((lens[:, None, None] + 127) // 128) * 128
        full = acc0 + ref_conv(xr, w, None, taps)
        assert rel_err(ya * must, full * must) < 1e-2
        assert torch.equal(ya * beyond, acc0 * beyond)
        between = (~must & ~beyond).expand_as(full)
        assert bool(((ya == acc0) | ((ya - full).abs() <= 1e-2 * full.abs().max()))[between].all())       # untouched, or accumulated
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('Cin,Cout,taps', [(256, 128, 3), (128, 256, 1), (128, 128, 3)])
def test_conv_tile_skipping_with_many_batch_rows(ops, Cin, Cout, taps):
    """More batch rows than the kernels number live tiles for (1024): the deep-K and weight-stationary kernels fall back to the
    plain tile order, and the bf16 weight gradient to per-row limits read from memory."""
    ops.set_precision('bf16')
    try:
        B, N = 1100, 130
        g = torch.Generator().manual_seed(5)
        lens = torch.randint(1, N + 1, (B,), generator=g).to(torch.int32).to(DEV)
        wshape = (Cout, Cin, 3) if taps == 3 else (Cout, Cin)
        w = randn(*wshape, seed=1, scale=1.0 / math.sqrt(Cin * taps))
        b = randn(Cout, seed=2, scale=0.1)
        x = randn(B, N, Cin, seed=3).to(torch.bfloat16)
        pack = ops.PackedWeight(w)
        ref = ref_conv(x.float(), w, b, taps)
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
        y = ops.conv_gemm(x, pack, b, lens=lens, mask_rows=True, halo=0)
        assert rel_err(y, ref * valid) < 1e-2 and torch.equal(y * ~valid, torch.zeros_like(y))
        dy = (randn(B, N, Cout, seed=4) * valid).to(torch.bfloat16)
        dw, db = ops.conv_wgrad(dy, x, pack, lens, 0)
        dw_ref, db_ref = ops.conv_wgrad(dy, x, pack)
        assert rel_err(dw, dw_ref) < 1e-4 and rel_err(db, db_ref) < 1e-4
    finally:
        ops.set_precision('f32')


def ref_attention(qkv, lens, heads, keep=None, p=0.0):
    B, N, D3 = qkv.shape
    D = D3 // 3
    q, k, v = qkv.split(D, dim=2)
    sp = lambda t: t.view(B, N, heads, 64).transpose(1, 2)
    q, k, v = sp(q) * 0.125, sp(k), sp(v)
    s = q @ k.transpose(-1, -2)
    pad = torch.arange(N, device=qkv.device)[None, :] >= lens[:, None]
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    ctx = (pr @ v).transpose(1, 2).reshape(B, N, D)
    valid = (~pad)[:, :, None].float()
    return ctx * valid


@pytest.mark.parametrize('precision,tol_f,tol_b', [('f32', 3e-6, 1e-5), ('bf16', 2e-2, 3e-2), ('fp16', 3e-3, 5e-3)])
@pytest.mark.parametrize('B,N,lens', [(2, 64, [64, 33]), (3, 150, [150, 149, 7]), (1, 257, [257])])
def test_attention_forward_backward(ops, B, N, lens, precision, tol_f, tol_b):
    ops.set_precision(precision)
    try:
        _attention_forward_backward(ops, B, N, lens, tol_f, tol_b)
    finally:
        ops.set_precision('f32')


def _attention_forward_backward(ops, B, N, lens, tol_f, tol_b):
    heads = 2
    qkv = randn(B, N, 384, seed=1).requires_grad_(True)
    ln = lens_tensor(lens)
    ctx, lse = ops.attention_fwd(qkv.detach(), ln, heads, 0, 0.0)
    ref = ref_attention(qkv, ln.long(), heads)
    assert rel_err(ctx, ref) < tol_f
    dctx = randn(B, N, 128, seed=2)
    valid = (torch.arange(N, device=DEV)[None, :] < ln[:, None])[:, :, None].float()
    dctx = dctx * valid                                    # the model never sends gradient into padded queries
    ref.backward(dctx)
    dqkv = ops.attention_bwd(qkv.detach(), ctx, dctx, lse, ln, heads, 0, 0.0)
    assert rel_err(dqkv, qkv.grad) < tol_b
    assert torch.isfinite(dqkv).all()
    if ops.get_precision() in ('bf16', 'fp16'):            # 16-bit-stored q/k/v and dqkv (what the FFT block uses in those modes)
        h16 = ops.hidden_dtype()
        qh = qkv.detach().to(h16)
        ctx_h, lse_h = ops.attention_fwd(qh, ln, heads, 0, 0.0)
        assert rel_err(ctx_h, ref.detach()) < tol_f
        dq_h = ops.attention_bwd(qh, ctx_h, dctx, lse_h, ln, heads, 0, 0.0, out_dtype=h16)
        assert dq_h.dtype == h16 and rel_err(dq_h.float(), qkv.grad) < tol_b
        # 16-bit-stored context as well (the FFT block's configuration): same results within the mode's tolerance
        ctx_c, lse_c = ops.attention_fwd(qh, ln, heads, 0, 0.0, ctx_dtype=h16)
        assert ctx_c.dtype == h16 and rel_err(ctx_c.float(), ref.detach()) < tol_f and torch.equal(lse_c, lse_h)
        dq_c = ops.attention_bwd(qh, ctx_c, dctx.to(h16), lse_c, ln, heads, 0, 0.0, out_dtype=h16)       # dctx is stored like ctx
        assert rel_err(dq_c.float(), qkv.grad) < tol_b


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
def test_attention_longest_first_order_changes_nothing(ops, precision):
    """``dx_length_order`` = stable argsort by length, descending; handing the attention workgroups out in that order is pure
    scheduling: forward, log-sum-exp and all three gradients are BITWISE the same with and without it, dropout included (the dropout
    counters are built from the utterance's own index, not from blockIdx)."""
    lens = [70, 200, 13, 200, 129, 64, 1]
    B, N, heads = len(lens), 200, 2
    ln = lens_tensor(lens)
    order = ops.length_order(ln)
    want = torch.tensor(sorted(range(B), key=lambda i: (-lens[i], i)), dtype=torch.int32)
    assert torch.equal(order.cpu(), want)
    ops.set_precision(precision)
    try:
        h = ops.hidden_dtype()
        qkv = randn(B, N, 384, seed=5).to(h)
        dctx = (randn(B, N, 128, seed=6) * (torch.arange(N, device=DEV)[None, :] < ln[:, None])[:, :, None].float()).to(h)
        for seed, p in ((0, 0.0), (77, 0.1)):
            a, la = ops.attention_fwd(qkv, ln, heads, seed, p, ctx_dtype=h)
            b, lb = ops.attention_fwd(qkv, ln, heads, seed, p, ctx_dtype=h, order=order)
            assert torch.equal(a, b) and torch.equal(la, lb)
            ga = ops.attention_bwd(qkv, a, dctx, la, ln, heads, seed, p, out_dtype=h)
            gb = ops.attention_bwd(qkv, b, dctx, lb, ln, heads, seed, p, out_dtype=h, order=order)
            assert torch.equal(ga, gb)
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['f32', 'bf16', 'fp16'])
def test_attention_dropout_mask_consistency(ops, precision):
    ops.set_precision(precision)
    try:
        _attention_dropout_mask_consistency(ops, 3e-6 if precision == 'f32' else 2e-2, 1e-5 if precision == 'f32' else 3e-2)
    finally:
        ops.set_precision('f32')


def _attention_dropout_mask_consistency(ops, tol_f, tol_b):
    """q = k = 0 and one-hot V make ctx[q][key] = keep[q,key] / (len (1-p)): the mask is observable, so the backward's
    regenerated mask can be checked against the forward's."""
    B, N, heads, p = 2, 64, 2, 0.25
    ln = lens_tensor([64, 48])
    qkv = torch.zeros(B, N, 384, device=DEV)
    eye = torch.eye(64, device=DEV)
    qkv[:, :, 256:320] = eye
    qkv[:, :, 320:384] = eye
    seed = 0xABCDEF1234
    ctx, lse = ops.attention_fwd(qkv, ln, heads, seed, p)
    keep = torch.zeros(B, heads, N, N, device=DEV)
    for b in range(B):
        n = int(ln[b])
        keep[b, 0, :, :] = (ctx[b, :, 0:64] * n * (1 - p)).round()
        keep[b, 1, :, :] = (ctx[b, :, 64:128] * n * (1 - p)).round()
    for b in range(B):
        n = int(ln[b])
        frac = keep[b, :, :n, :n].mean().item()
        assert abs(frac - (1 - p)) < 0.03, frac
    ctx2, _ = ops.attention_fwd(qkv, ln, heads, seed, p)
    assert torch.equal(ctx, ctx2)                          # deterministic in (seed, index)
    ctx3, _ = ops.attention_fwd(qkv, ln, heads, seed + 1, p)
    assert not torch.equal(ctx, ctx3)
    # full backward with random q/k/v against autograd using the recovered mask
    qkv_r = randn(B, N, 384, seed=5).requires_grad_(True)
    ctx_r, lse_r = ops.attention_fwd(qkv_r.detach(), ln, heads, seed, p)
    ref = ref_attention(qkv_r, ln.long(), heads, keep=keep, p=p)
    assert rel_err(ctx_r, ref) < tol_f
    valid = (torch.arange(N, device=DEV)[None, :] < ln[:, None])[:, :, None].float()
    dctx = randn(B, N, 128, seed=6) * valid
    ref.backward(dctx)
    dqkv = ops.attention_bwd(qkv_r.detach(), ctx_r, dctx, lse_r, ln, heads, seed, p)
    assert rel_err(dqkv, qkv_r.grad) < tol_b


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('p_drop', [0.0, 0.2])
@pytest.mark.parametrize('rep', [1, 7])      # 5 / 35 utterances: 62-token tiles (short batches) / 126-token tiles (>= 96 of them)
def test_ff_pair_backward_layernorm_epilogue_equals_two_launches(ops, precision, p_drop, rep):
    """dx_ff_pair_lnbwd == dx_ff_pair(backward, accumulate) followed by dx_ln_bwd (same seeds): dz1, its dropped-out 16-bit copy, dh and the
    affine gradients; lengths on a tile edge, on the halo row and inside a tile."""
    ops.set_precision(precision)
    try:
        h16 = ops.hidden_dtype()
        B, N, Fc = 5 * rep, 300, 1024
        lens = lens_tensor([300, 252, 126, 127, 40] * rep)
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
        vf = valid[:, :, None].float()
        w1 = randn(Fc, 128, 3, seed=2, scale=1 / math.sqrt(384))
        w2 = randn(128, Fc, 3, seed=4, scale=1 / math.sqrt(3 * Fc))
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        x = (randn(B, N, 128, seed=1) * vf).to(h16)
        h = ops.conv_gemm(x, p1, randn(Fc, seed=3, scale=0.1), relu=True, lens=lens, halo=1, out_dtype=h16)     # the stored forward activation (mask)
        dff = (randn(B, N, 128, seed=5) * vf).to(h16)
        resid = randn(B, N, 128, seed=6) * vf                                # the residual-branch gradient already in the output buffer
        z = randn(B, N, 128, seed=7)
        mean, rstd = z.mean(dim=2), 1.0 / torch.sqrt(z.var(dim=2, unbiased=False) + 1e-5)
        lw, lb = 1 + 0.1 * randn(128, seed=8), randn(128, seed=9, scale=0.1)
        out0 = resid.clone()
        dy1, dh0 = ops.ff_pair(dff, p1, p2, None, None, lens, backward=True, aux=h, out=out0, accumulate=True)
        r = ops.ln_bwd(dy1, z, mean, rstd, lw, lb, None, lens, want_da=p_drop > 0, seed_pre=55, p_pre=p_drop, shadow=True)
        dz0, dw0, db0, dg0 = r[0], r[2], r[3], r[5]
        out1 = resid.clone()
        dz1, dh1, dg1, dw1, db1 = ops.ff_pair_lnbwd(dff, p1, p2, lens, h, out1, z, mean, rstd, lw, lb, seed_pre=55, p_pre=p_drop)
        assert dz1.data_ptr() == out1.data_ptr() and torch.equal(dh0, dh1)
        assert rel_err(dz1, dz0) < 1e-5 and float(dz1[~valid].abs().max()) == 0.0
        assert rel_err(dg1.float(), dg0.float()) < 1e-2 and float(dg1[~valid].float().abs().max()) == 0.0
        assert rel_err(dw1, dw0) < 1e-4 and rel_err(db1, db0) < 1e-4
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('p_drop,use_film', [(0.0, True), (0.2, True), (0.2, False)])
@pytest.mark.parametrize('rep', [1, 7])      # 5 / 35 utterances: 62-token tiles (short batches) / 126-token tiles (>= 96 of them)
def test_ff_block_backward_equals_three_launches(ops, precision, p_drop, use_film, rep):
    """dx_ff_block_bwd == dx_ln_bwd (second LayerNorm) + dx_ff_pair (input-gradient pair, accumulate) + dx_ln_bwd (first LayerNorm), same
    seeds: dz1, both 16-bit gradient copies, the hidden gradient, the affine and FiLM gradients; lengths on tile edges and halo rows."""
    ops.set_precision(precision)
    try:
        h16 = ops.hidden_dtype()
        B, N, Fc = 5 * rep, 300, 1024
        lens = lens_tensor([300, 252, 126, 127, 40] * rep)
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
        vf = valid[:, :, None].float()
        w1 = randn(Fc, 128, 3, seed=2, scale=1 / math.sqrt(384))
        w2 = randn(128, Fc, 3, seed=4, scale=1 / math.sqrt(3 * Fc))
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        x = (randn(B, N, 128, seed=1) * vf).to(h16)
        h = ops.conv_gemm(x, p1, randn(Fc, seed=3, scale=0.1), relu=True, lens=lens, halo=1, out_dtype=h16)
        dy2 = randn(B, N, 128, seed=5) * vf
        z2, z1 = randn(B, N, 128, seed=6), randn(B, N, 128, seed=7)
        st = lambda z: (z.mean(dim=2), 1.0 / torch.sqrt(z.var(dim=2, unbiased=False) + 1e-5))
        (m2, r2), (m1, r1) = st(z2), st(z1)
        l2w, l2b = 1 + 0.1 * randn(128, seed=8), randn(128, seed=9, scale=0.1)
        l1w, l1b = 1 + 0.1 * randn(128, seed=10), randn(128, seed=11, scale=0.1)
        film = randn(B, 256, seed=12) if use_film else None
        # three launches
        a = ops.ln_bwd(dy2, z2, m2, r2, l2w, l2b, film, lens, want_da=p_drop > 0, seed_pre=71, p_pre=p_drop, shadow=True)
        dz2, dw2, db2, dfilm0, dff0 = a[0], a[2], a[3], a[4], a[5]
        dy1, dh0 = ops.ff_pair(dff0, p1, p2, None, None, lens, backward=True, aux=h, out=dz2, accumulate=True)
        b_ = ops.ln_bwd(dy1, z1, m1, r1, l1w, l1b, None, lens, want_da=p_drop > 0, seed_pre=72, p_pre=p_drop, shadow=True)
        dz1_0, dw1_0, db1_0, dg1_0 = b_[0], b_[2], b_[3], b_[5]
        # one launch
        wo = randn(128, 128, seed=13, scale=0.09)
        po = ops.PackedWeight(wo)
        dz1, dh, dg1, dg2, dfilm, dw2n, db2n, dw1n, db1n, datt = ops.ff_block_bwd(dy2, z2, m2, r2, l2w, l2b, film, p1, p2, lens, h, z1, m1, r1, l1w, l1b,
                                                                                  seed2=71, p2=p_drop, seed1=72, p1=p_drop, out_pack=po)
        datt0 = ops.conv_gemm(dg1, po, None, transpose=True, lens=lens, halo=0, out_dtype=h16)      # the launch it replaces, on the same operand
        assert rel_err(datt.float(), datt0.float()) < 1e-2 and float(datt[~valid].float().abs().max()) == 0.0
        # the prologue sums a row's channels in another order than dx_ln_bwd: equal up to one 16-bit rounding step of the copy
        assert rel_err(dg2.float(), dff0.float()) < 1e-2 and float(dg2[~valid].float().abs().max()) == 0.0
        assert rel_err(dh.float(), dh0.float()) < 2e-2
        assert rel_err(dz1, dz1_0) < 1e-3 and float(dz1[~valid].abs().max()) == 0.0       # (downstream of the 16-bit copy's rounding step)
        assert rel_err(dg1.float(), dg1_0.float()) < 1e-2 and float(dg1[~valid].float().abs().max()) == 0.0
        assert rel_err(dw2n, dw2) < 1e-4 and rel_err(db2n, db2) < 1e-4 and rel_err(dw1n, dw1_0) < 1e-3 and rel_err(db1n, db1_0) < 1e-3
        if use_film:
            assert rel_err(dfilm, dfilm0) < 1e-4
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('rep', [1, 7])      # 5 / 35 utterances: 62-token tiles (short batches) / 126-token tiles (>= 96 of them)
def test_ff_block_backward_with_sign_words_equals_the_stored_activation(ops, precision, rep):
    """The backward's ReLU mask from the forward's sign words (``hmask``: one bit per hidden element in the kernel's register layout) is the
    mask ``h > 0`` it replaces: same hidden gradient, same dz1, bit for bit; utterances on tile edges, on the halo row, inside a tile and
    longer than one tile."""
    ops.set_precision(precision)
    try:
        h16 = ops.hidden_dtype()
        B, N, Fc = 6 * rep, 300, 1024
        lens = lens_tensor([300, 252, 126, 127, 40, 1] * rep)
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
        vf = valid[:, :, None].float()
        w1 = randn(Fc, 128, 3, seed=2, scale=1 / math.sqrt(384))
        w2 = randn(128, Fc, 3, seed=4, scale=1 / math.sqrt(3 * Fc))
        b1, b2 = randn(Fc, seed=3, scale=0.1), randn(128, seed=33, scale=0.1)
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        x = (randn(B, N, 128, seed=1) * vf).to(h16)
        res = randn(B, N, 128, seed=14) * vf
        l2w, l2b = 1 + 0.1 * randn(128, seed=8), randn(128, seed=9, scale=0.1)
        l1w, l1b = 1 + 0.1 * randn(128, seed=10), randn(128, seed=11, scale=0.1)
        film = randn(B, 256, seed=12)
        z2, h, _, m2, r2, mask = ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, l2w, l2b, film, want_mask=True)
        assert mask is not None and mask.dtype == torch.int32
        dy2 = randn(B, N, 128, seed=5) * vf
        z1 = randn(B, N, 128, seed=7)
        m1, r1 = z1.mean(dim=2), 1.0 / torch.sqrt(z1.var(dim=2, unbiased=False) + 1e-5)
        po = ops.PackedWeight(randn(128, 128, seed=13, scale=0.09))
        kw = dict(seed2=71, p2=0.0, seed1=72, p1=0.0, out_pack=po)
        a = ops.ff_block_bwd(dy2, z2, m2, r2, l2w, l2b, film, p1, p2, lens, h, z1, m1, r1, l1w, l1b, **kw)
        b = ops.ff_block_bwd(dy2, z2, m2, r2, l2w, l2b, film, p1, p2, lens, h, z1, m1, r1, l1w, l1b, hmask=mask, **kw)
        assert torch.equal(a[1], b[1]), float((a[1].float() - b[1].float()).abs().max())     # the hidden gradient
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[9], b[9])
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['f32', 'bf16', 'fp16'])
def test_channel_affine_vs_torch(ops, precision):
    """dx_channel_affine (eval-mode BatchNorm of the frozen pitch predictor, layers/pitch_predictor.py:49-62) in fp32 and in the 16-bit
    storage of the two reduced operand modes: the result is the fp32 affine of the stored value, rounded once."""
    ops.set_precision(precision)
    try:
        x = randn(3, 217, 256, seed=4).to(ops.hidden_dtype())
        sc, sh = randn(256, seed=5), randn(256, seed=6)
        y = ops.channel_affine(x, sc, sh)
        ref = x.float() * sc + sh
        ulp = {'f32': 2.0 ** -23, 'bf16': 2.0 ** -8, 'fp16': 2.0 ** -11}[precision]     # one rounding of the stored result (+ fma contraction)
        assert y.dtype == x.dtype and bool(((y.float() - ref).abs() <= 1.01 * ulp * ref.abs() + 1e-6).all())
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('with_pm', [True, False])
def test_film_affine_forward_backward_vs_torch(ops, with_pm):
    """FilmAffineFn (StyleAdapter tail, model.py:779-800) against the element-wise torch formulation, incl. blocks that receive no gradient."""
    from ubisoft_laforge_daft_exprt_amd import functional as Fx
    B, nb, C = 5, 8, 128
    gam = randn(B, nb * C, seed=1).requires_grad_(True)
    bet = randn(B, nb * C, seed=2).requires_grad_(True)
    pm = randn(2, nb, seed=3).requires_grad_(True) if with_pm else None
    *blocks, whole = Fx.FilmAffineFn.apply(gam, bet, pm, nb, None)
    g = gam.view(B, nb, C)
    b = bet.view(B, nb, C)
    ref = torch.cat((pm[0][None, :, None] * g + 1, pm[1][None, :, None] * b), dim=2) if with_pm else torch.cat((g + 1, b), dim=2)
    for i in range(nb):
        assert blocks[i].is_contiguous() and rel_err(blocks[i], ref[:, i]) < 1e-6
    assert rel_err(whole.transpose(0, 1), ref) < 1e-6
    w = [randn(B, 2 * C, seed=10 + i) for i in range(nb)]
    used = [0, 1, 3, 4, 5, 7]                                  # blocks 2 and 6 get no gradient
    loss = sum((blocks[i] * w[i]).sum() for i in used)
    loss.backward()
    got = (gam.grad.clone(), bet.grad.clone(), None if pm is None else pm.grad.clone())
    gam.grad = bet.grad = None
    if pm is not None:
        pm.grad = None
    sum((ref[:, i] * w[i]).sum() for i in used).backward()
    assert rel_err(got[0], gam.grad) < 1e-6 and rel_err(got[1], bet.grad) < 1e-6
    if with_pm:
        assert rel_err(got[2], pm.grad) < 1e-5


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('p_drop', [0.0, 0.2])
@pytest.mark.parametrize('rep', [1, 7])      # 5 / 35 utterances: 62-token tiles (short batches) / 126-token tiles (>= 96 of them)
def test_ff_pair_layernorm_epilogue_equals_two_launches(ops, precision, p_drop, rep):
    """dx_ff_pair_ln == dx_ff_pair followed by dx_ln_fwd (same seeds): z, statistics, y, the hidden tensor; lengths that put the last tile on
    the halo row, on a tile edge and inside a tile; FiLM on and off."""
    ops.set_precision(precision)
    try:
        h16 = ops.hidden_dtype()
        B, N, Fc = 5 * rep, 300, 1024
        lens = lens_tensor([300, 252, 126, 127, 40] * rep)
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
        x = (randn(B, N, 128, seed=1) * valid[:, :, None]).to(h16)
        w1, b1 = randn(Fc, 128, 3, seed=2, scale=1 / math.sqrt(384)), randn(Fc, seed=3, scale=0.1)
        w2, b2 = randn(128, Fc, 3, seed=4, scale=1 / math.sqrt(3 * Fc)), randn(128, seed=5, scale=0.1)
        res = randn(B, N, 128, seed=6)
        lw, lb = 1 + 0.1 * randn(128, seed=7), randn(128, seed=8, scale=0.1)
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        for film in (None, randn(B, 256, seed=9)):
            z0, h0 = ops.ff_pair(x, p1, p2, b1, b2, lens)
            y0, m0, r0 = ops.ln_fwd(z0, res, lw, lb, film, lens, seed_pre=91, p_pre=p_drop)
            z1, h1, y1, m1, r1 = ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, lw, lb, film, seed_pre=91, p_pre=p_drop)
            assert torch.equal(h0, h1)
            assert rel_err(z1[valid], z0[valid]) < 1e-6                 # z0 was overwritten with z by ln_fwd
            assert rel_err(m1[valid], m0[valid]) < 1e-5 and rel_err(r1[valid], r0[valid]) < 1e-4
            assert rel_err(y1, y0) < 1e-4 and float(y1[~valid].abs().max()) == 0.0
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('rep', [1, 7])      # 5 / 35 utterances: 62-token tiles (short batches) / 126-token tiles (>= 96 of them)
def test_ff_pair_epilogue_produces_next_blocks_qkv(ops, precision, rep):
    """dx_ff_pair_ln_qkv == dx_ff_pair_ln followed by the next block's in-projection launch (dx_conv_gemm 128 -> 384, halo 0) on its y:
    everything dx_ff_pair_ln returns is unchanged bit for bit; qkv agrees to the rounding of its 16-bit storage (other fp32 summation
    order), equals the bias on padded rows of live tiles and zero in tiles beyond the halo."""
    ops.set_precision(precision)
    try:
        h16 = ops.hidden_dtype()
        B, N, Fc = 5 * rep, 300, 1024
        lens_l = [300, 252, 126, 127, 40] * rep
        lens = lens_tensor(lens_l)
        valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
        x = (randn(B, N, 128, seed=1) * valid[:, :, None]).to(h16)
        w1, b1 = randn(Fc, 128, 3, seed=2, scale=1 / math.sqrt(384)), randn(Fc, seed=3, scale=0.1)
        w2, b2 = randn(128, Fc, 3, seed=4, scale=1 / math.sqrt(3 * Fc)), randn(128, seed=5, scale=0.1)
        res = randn(B, N, 128, seed=6)
        lw, lb = 1 + 0.1 * randn(128, seed=7), randn(128, seed=8, scale=0.1)
        wq, bq = randn(384, 128, seed=10, scale=1 / math.sqrt(128)), randn(384, seed=11, scale=0.1)
        p1, p2, pq = ops.PackedWeight(w1), ops.PackedWeight(w2), ops.PackedWeight(wq)
        assert ops.next_qkv_applies(pq, precision)
        for film in (None, randn(B, 256, seed=9)):
            z0, h0, y0, m0, r0 = ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, lw, lb, film, seed_pre=91, p_pre=0.1)
            q0 = ops.conv_gemm(y0, pq, bq, lens=lens, halo=0, out_dtype=h16)
            z1, h1, y1, m1, r1, q1 = ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, lw, lb, film, seed_pre=91, p_pre=0.1, next_in=(pq, bq))
            assert torch.equal(h0, h1) and torch.equal(y0, y1)
            for a, b in ((z0, z1), (m0, m1), (r0, r1)):              # (tiles beyond the halo leave these three unwritten)
                assert torch.equal(a[valid], b[valid])
            assert q1.dtype == h16 and q1.shape == (B, N, 384)
            # forward-only form (inference): no hidden tensor, everything else bit for bit
            z2, h2, y2, m2, r2, q2 = ops.ff_pair_ln(x, p1, p2, b1, b2, lens, res, lw, lb, film, seed_pre=91, p_pre=0.1, next_in=(pq, bq), need_h=False)
            assert h2 is None and torch.equal(y2, y1) and torch.equal(q2, q1) and torch.equal(z2[valid], z1[valid]) and torch.equal(m2[valid], m1[valid])
            ulp = 2.0 ** -7 if precision == 'bf16' else 2.0 ** -10
            diff = (q1.float() - q0.float()).abs()[valid]              # (padded rows: the two launches tile the axis differently, see below)
            assert float((diff / q0.float()[valid].abs().clamp_min(0.25)).max()) <= 2 * ulp   # at most a rounding step of the 16-bit result
            assert float((diff > 0).float().mean()) < 0.02                                        # and only where the fp32 sums straddle one
            tok = 126 if B * -(-N // 126) >= 96 else 62                                           # the tile width the launch chose for this shape
            for b, n in enumerate(lens_l):
                live_end = min(N, -(-(n + 1) // tok) * tok)                                      # the tiles that start before len + halo (1)
                if n < live_end:
                    assert torch.equal(q1[b, n:live_end].float(), bq.to(h16).float().expand(live_end - n, 384))
                assert float(q1[b, live_end:].float().abs().max() if live_end < N else 0.0) == 0.0
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('p_drop', [0.0, 0.2])
def test_fused_out_projection_layernorm_equals_two_launches(ops, precision, p_drop):
    """dx_proj_ln_fwd == dx_conv_gemm (128 -> 128) followed by dx_ln_fwd, same seeds: z / mean / rstd / y / 16-bit copy, padded rows,
    a row count that is not a multiple of the 64-row tile, FiLM on and off."""
    ops.set_precision(precision)
    try:
        h16 = ops.hidden_dtype()
        B, N = 3, 150
        lens = lens_tensor([150, 97, 3])
        x = randn(B, N, 128, seed=1).to(h16)
        w = randn(128, 128, seed=2, scale=0.09)
        pb = randn(128, seed=3, scale=0.1)
        res = randn(B, N, 128, seed=4)
        lw, lb = 1 + 0.1 * randn(128, seed=5), randn(128, seed=6, scale=0.1)
        pack = ops.PackedWeight(w)
        for film in (None, randn(B, 256, seed=7)):
            a = ops.conv_gemm(x, pack, pb, lens=lens, halo=0)
            y0, m0, r0, yh0 = ops.ln_fwd(a, res, lw, lb, film, lens, seed_pre=77, p_pre=p_drop, shadow=True)
            z1, y1, m1, r1, yh1 = ops.proj_ln_fwd(x, pack, pb, res, lw, lb, film, lens, seed_pre=77, p_pre=p_drop, shadow=True)
            valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
            assert rel_err(z1[valid], a[valid]) < 1e-5              # `a` now holds z (in place)
            assert rel_err(m1, m0) < 1e-5 and rel_err(r1, r0) < 1e-4
            assert rel_err(y1, y0) < 1e-4 and float(y1[~valid].abs().max()) == 0.0
            assert rel_err(yh1.float(), yh0.float()) < 1e-2
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('C', [128, 1024])
@pytest.mark.parametrize('use_film,use_res,use_mask', [(True, True, True), (False, True, True), (False, False, False)])
def test_layernorm_family(ops, C, use_film, use_res, use_mask):
    if use_film and C != 128:
        pytest.skip('FiLM is only used at C=128')
    B, N = 3, 45
    a = randn(B, N, C, seed=1).requires_grad_(True)
    res = randn(B, N, C, seed=2).requires_grad_(True) if use_res else None
    w = (1 + 0.1 * randn(C, seed=3)).requires_grad_(True)
    b = randn(C, seed=4, scale=0.1).requires_grad_(True)
    film = randn(B, 2 * C, seed=5).requires_grad_(True) if use_film else None
    lens = lens_tensor([45, 44, 9]) if use_mask else None
    z = a if res is None else a + res
    ref = F.layer_norm(z, (C,), w, b, 1e-5)
    if film is not None:
        ref = film[:, None, :C] * ref + film[:, None, C:]
    if lens is not None:
        ref = ref * (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
    a_k = a.detach().clone()
    y, mean, rstd = ops.ln_fwd(a_k, None if res is None else res.detach(), w.detach(), b.detach(),
                               None if film is None else film.detach(), lens)
    assert rel_err(y, ref) < 3e-6
    vmask = torch.ones(B, N, 1, device=DEV) if lens is None else (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None].float()
    assert rel_err(a_k * vmask, z.detach() * vmask) < 1e-7   # z written back in place (valid rows; padded rows are not computed)
    dy = randn(B, N, C, seed=6)
    ref.backward(dy)
    dz, da, dw, db, dfilm = ops.ln_bwd(dy, a_k, mean, rstd, w.detach(), b.detach(), None if film is None else film.detach(), lens)
    assert rel_err(dz, a.grad) < 1e-5
    assert rel_err(dw, w.grad) < 1e-5 and rel_err(db, b.grad) < 1e-5
    if film is not None:
        assert rel_err(dfilm, film.grad) < 1e-5


def test_layernorm_dropout_and_relu_mask(ops):
    B, N, C, p = 2, 33, 128, 0.2
    a0 = randn(B, N, C, seed=1)
    res = randn(B, N, C, seed=2)
    w, b = 1 + 0.1 * randn(C, seed=3), randn(C, seed=4, scale=0.1)
    lens = lens_tensor([33, 20])
    a_k = a0.clone()
    y, mean, rstd = ops.ln_fwd(a_k, res, w, b, None, lens, seed_pre=77, p_pre=p)
    vmask = (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
    keep = torch.where(vmask, ((a_k - res) / a0 * (1 - p)).round(), torch.ones_like(a0))   # observable pre-dropout mask (valid rows)
    assert set(keep.unique().tolist()) <= {0.0, 1.0}
    assert abs(keep[vmask.expand_as(keep)].mean().item() - (1 - p)) < 0.03
    a = a0.clone().requires_grad_(True)
    ref = F.layer_norm(a * keep / (1 - p) + res, (C,), w, b, 1e-5) * (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
    assert rel_err(y, ref) < 3e-6
    dy = randn(B, N, C, seed=5)
    ref.backward(dy)
    dz, da, dw, db, _ = ops.ln_bwd(dy, a_k, mean, rstd, w, b, None, lens, want_da=True, seed_pre=77, p_pre=p)
    assert rel_err(da, a.grad) < 1e-5
    # prenet form: ReLU'd input, LayerNorm, dropout on the OUTPUT, no mask
    x = F.relu(randn(B, N, 1024, seed=7)).requires_grad_(True)
    w2, b2 = 1 + 0.1 * randn(1024, seed=8), randn(1024, seed=9, scale=0.1)
    xk = x.detach().clone()
    y2, m2, r2 = ops.ln_fwd(xk, None, w2, b2, None, None, seed_post=5, p_post=p)
    ln = F.layer_norm(x, (1024,), w2, b2, 1e-5)
    keep2 = (y2 / ln.detach() * (1 - p)).round()
    assert abs(keep2.mean().item() - (1 - p)) < 0.02
    ref2 = ln * keep2 / (1 - p)
    dy2 = randn(B, N, 1024, seed=10)
    ref2.backward(dy2)
    dz2, _, _, _, _ = ops.ln_bwd(dy2, xk, m2, r2, w2, b2, None, None, relu_mask=True, seed_post=5, p_post=p)
    assert rel_err(dz2, x.grad * (x.detach() > 0)) < 1e-5


def test_layernorm_1024_bf16_rows(ops):
    """bf16-stored 1024-wide rows (prenet, bf16 mode): ReLU'd input -> LayerNorm -> dropout, halo skipping, and the backward."""
    B, N, C, p = 3, 70, 1024, 0.1
    lens = lens_tensor([70, 40, 12])
    x = F.relu(randn(B, N, C, seed=1)).to(torch.bfloat16)
    w, b = 1 + 0.1 * randn(C, seed=2), randn(C, seed=3, scale=0.1)
    xk = x.clone()
    y, mean, rstd = ops.ln_fwd(xk, None, w, b, None, lens, seed_post=9, p_post=p, halo=2)
    assert y.dtype == torch.bfloat16 and torch.equal(xk, x)
    xf = x.float().requires_grad_(True)
    ln = F.layer_norm(xf, (C,), w, b, 1e-5)
    live = (torch.arange(N, device=DEV)[None, :] < (lens + 2)[:, None])[:, :, None]
    keep = torch.where(live, (y.float() / ln.detach() * (1 - p)).round().clamp(0, 1), torch.ones_like(ln))
    assert abs(keep[live.expand_as(keep)].mean().item() - (1 - p)) < 0.02
    ref = ln * keep / (1 - p) * live
    assert rel_err(y.float(), ref.detach()) < 1e-2 and torch.equal(y.float() * ~live, torch.zeros_like(ref))
    dy = (randn(B, N, C, seed=4) * live).to(torch.bfloat16)
    ref.backward(dy.float())
    dz, _, dw, db, _ = ops.ln_bwd(dy, xk, mean, rstd, w, b, None, lens, relu_mask=True, seed_post=9, p_post=p, halo=2)
    assert dz.dtype == torch.bfloat16
    assert rel_err(dz.float(), xf.grad * (x.float() > 0)) < 2e-2


def test_embedding_positions_masks_pool(ops):
    from oracle import daft_exprt_oracle as oracle
    B, N, D = 3, 21, 128
    lens = lens_tensor([21, 20, 5])
    pe = oracle.positional_table(D).to(DEV)
    emb = randn(76, D, seed=1)
    sym = torch.randint(1, 76, (B, N), generator=torch.Generator().manual_seed(2)).to(DEV)
    valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
    out = ops.add_pos(None, sym, emb, pe, lens)
    ref = (emb[sym] + pe[:N][None]) * valid
    assert torch.equal(out, ref)
    x = randn(B, N, D, seed=3)
    assert torch.equal(ops.add_pos(x, None, None, pe, lens), (x + pe[:N][None]) * valid)
    assert torch.equal(ops.mask_rows(x, lens), x * valid)
    demb = ops.embedding_bwd(x, sym, lens, 76)
    ref_d = torch.zeros(76, D, device=DEV).index_add_(0, sym[valid[:, :, 0]], x[valid[:, :, 0]])
    assert rel_err(demb, ref_d) < 1e-6
    pooled = ops.mean_pool(x * valid, lens)
    assert rel_err(pooled, (x * valid).sum(1) / lens[:, None]) < 1e-6
    dp = randn(B, D, seed=4)
    assert rel_err(ops.mean_pool_bwd(dp, lens, N), (dp / lens[:, None])[:, None, :] * valid) < 1e-6
    t = randn(2, 37, 80, seed=5)
    assert torch.equal(ops.transpose(t), t.transpose(1, 2).contiguous())
    e = randn(4, 192, seed=6)
    assert rel_err(ops.l2_normalize(e), F.normalize(e, p=2, dim=-1)) < 1e-6
    logits = randn(5, 3, seed=7).requires_grad_(True)
    tgt = torch.tensor([0, 2, 1, 1, 0], device=DEV)
    loss, dl = ops.cross_entropy(logits.detach(), tgt)
    ref_l = F.cross_entropy(logits, tgt)
    ref_l.backward()
    assert abs(loss.item() - ref_l.item()) < 1e-6 and rel_err(dl, logits.grad) < 1e-5


def test_accent_sum_and_scalar_conv_grads(ops):
    from oracle import daft_exprt_oracle as oracle
    B, N, D = 2, 40, 128
    lens = lens_tensor([40, 38])
    pe = oracle.positional_table(D).to(DEV)
    prenet = randn(B, N, D, seed=1)
    energy, pitch = randn(B, N, seed=2), randn(B, N, seed=3)
    we, be = randn(D, 1, 3, seed=4).requires_grad_(True), randn(D, seed=5).requires_grad_(True)
    wp, bp = randn(D, 1, 3, seed=6).requires_grad_(True), randn(D, seed=7).requires_grad_(True)
    valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])[:, :, None]
    conv = lambda s, w, b: F.conv1d(s[:, None, :], w, b, padding=1).transpose(1, 2)
    ref = (prenet + conv(energy, we, be) + conv(pitch, wp, bp) + pe[:N][None]) * valid
    out = ops.accent_sum(prenet, energy, pitch, we.detach(), be.detach(), wp.detach(), bp.detach(), pe, lens)
    assert rel_err(out, ref) < 1e-6
    dout = randn(B, N, D, seed=8) * valid
    ref.backward(dout)
    dw0, db0, dw1, db1 = ops.scalar_conv_wgrad(dout, energy, pitch, lens)
    assert rel_err(dw0, we.grad) < 1e-5 and rel_err(db0, be.grad) < 1e-5
    assert rel_err(dw1, wp.grad) < 1e-5 and rel_err(db1, bp.grad) < 1e-5


@pytest.mark.parametrize('B,L,lens,dur_hi', [(3, 14, [14, 13, 6], 7), (2, 300, [300, 211], 12)])
def test_gaussian_upsampler(ops, B, L, lens, dur_hi):
    from oracle import daft_exprt_oracle as oracle
    g = torch.Generator().manual_seed(3)
    lens_cpu = torch.tensor(lens)
    valid = torch.arange(L)[None, :] < lens_cpu[:, None]
    dur_int = torch.randint(0, dur_hi, (B, L), generator=g) * valid
    dur_int[:, 0] = 3
    dur = dur_int.float() * (256 / 22050)
    energy = torch.randn(B, L, generator=g) * valid
    pitch = torch.randn(B, L, generator=g) * valid
    D = 128
    names = ['duration_projection', 'energy_projection', 'pitch_projection']
    sd = {}
    for i, n in enumerate(names):
        sd[f'gaussian_upsampling.{n}.conv.weight'] = (0.5 * torch.randn(D, 1, 3, generator=g)).requires_grad_(True)
        sd[f'gaussian_upsampling.{n}.conv.bias'] = (0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    sd['gaussian_upsampling.projection.0.linear_layer.weight'] = (0.1 * torch.randn(1, D, generator=g)).requires_grad_(True)
    sd['gaussian_upsampling.projection.0.linear_layer.bias'] = torch.tensor([0.3]).requires_grad_(True)
    enc = (torch.randn(B, L, D, generator=g) * valid[:, :, None]).requires_grad_(True)
    xup_ref, w_ref = oracle.gaussian_upsampling(sd, enc, dur, dur_int, energy, pitch, lens_cpu)
    dev = lambda t: t.detach().to(DEV)
    gp = 'gaussian_upsampling.'
    lens_d = lens_tensor(lens)
    args = [dev(sd[gp + f'{n}.conv.{k}']) for n in names for k in ('weight', 'bias')]
    wr, br = dev(sd[gp + 'projection.0.linear_layer.weight']), dev(sd[gp + 'projection.0.linear_layer.bias'])
    xs, z, sigma = ops.upsample_prep(dev(enc), dev(dur), dev(energy), dev(pitch), *args, wr, br, lens_d)
    mu, totals = ops.duration_scan(dur_int.to(DEV))
    assert torch.equal(totals.cpu(), dur_int.sum(1))                      # integer path: exact
    T = int(totals.max())
    assert T == xup_ref.shape[1]
    xup, w = ops.upsample_fwd(xs, mu, sigma, lens_d, T)
    assert (w.cpu() - w_ref).abs().max() < 1e-5
    assert rel_err(xup.cpu(), xup_ref.detach()) < 5e-6
    dx = torch.randn(B, T, D, generator=g)
    xup_ref.backward(dx)
    dxs, dsigma = ops.upsample_bwd(dx.to(DEV), xs, mu, sigma, w, lens_d)
    dxs_tot, dz, dwr, dbr = ops.upsample_sym_bwd(dxs, dsigma, xs, z, dev(dur), lens_d, args[0], args[1], wr)
    assert rel_err(dxs_tot.cpu(), enc.grad) < 2e-4
    assert rel_err(dwr.cpu(), sd[gp + 'projection.0.linear_layer.weight'].grad[0]) < 2e-4
    assert rel_err(dbr.cpu(), sd[gp + 'projection.0.linear_layer.bias'].grad) < 2e-4
    dwe, dbe, dwp, dbp = ops.scalar_conv_wgrad(dxs_tot, dev(energy), dev(pitch), lens_d)
    assert rel_err(dwe.cpu(), sd[gp + 'energy_projection.conv.weight'].grad) < 2e-4
    assert rel_err(dwp.cpu(), sd[gp + 'pitch_projection.conv.weight'].grad) < 2e-4
    assert rel_err(dbe.cpu(), sd[gp + 'energy_projection.conv.bias'].grad) < 2e-4
    dwd, dbd, _, _ = ops.scalar_conv_wgrad(wr.view(D), dev(dur), None, lens_d, rowscale=dz)
    assert rel_err(dwd.cpu(), sd[gp + 'duration_projection.conv.weight'].grad) < 2e-4
    assert rel_err(dbd.cpu(), sd[gp + 'duration_projection.conv.bias'].grad) < 2e-4


def test_loss_kernels(ops):
    B, M, T = 3, 80, 70
    lens = lens_tensor([70, 69, 31])
    valid = (torch.arange(T, device=DEV)[None, :] < lens[:, None])
    mp = (randn(B, M, T, seed=1) * valid[:, None, :]).requires_grad_(True)
    mt = (-5 + 2 * randn(B, M, T, seed=2)).clamp(-11.5, 2) * valid[:, None, :]
    denom = M * lens.float()
    l1 = ((mp - mt).abs().sum((1, 2)) / denom).mean()
    l2 = (((mp - mt) ** 2).sum((1, 2)) / denom).mean()
    pe_ = F.avg_pool1d(torch.norm(torch.exp(mp), dim=1)[:, None], 5, 1, 2)[:, 0]
    te_ = F.avg_pool1d(torch.norm(torch.exp(mt), dim=1)[:, None], 5, 1, 2)[:, 0]
    en = (((pe_ - te_) ** 2) * valid).sum() / lens.sum().float()
    total = 1.0 * (l1 + l2) + 0.05 * en
    total.backward()
    ep, et, sums = ops.mel_stats(mp.detach(), mt)
    assert abs((sums[0] / denom).mean().item() - l1.item()) < 1e-5 * max(1, l1.item())
    assert abs((sums[1] / denom).mean().item() - l2.item()) < 1e-5 * max(1, l2.item())
    des, esum = ops.energy_diff(ep, et, lens)
    assert abs(esum.item() / lens.sum().item() - en.item()) < 1e-5 * max(1, en.item())
    dmel = ops.mel_grad(mp.detach(), mt, ep, des, lens, 1.0 / (M * B), 1.0 / (M * B), 0.05 / lens.sum().item())
    assert rel_err(dmel, mp.grad) < 1e-5
    pp = randn(B, T, seed=3).requires_grad_(True)
    gt = randn(B, T, seed=4) * (randn(B, T, seed=5) > -0.5)
    mask = (valid & (gt != 0)).float()
    pl = (((pp - gt) ** 2) * mask).sum() / (mask.sum() + 1e-5)
    (0.15 * pl).backward()
    s = ops.pitch_mse(pp.detach(), gt, lens)
    assert abs(s[0].item() / (s[1].item() + 1e-5) - pl.item()) < 1e-5 * max(1, pl.item())
    assert rel_err(ops.pitch_grad(pp.detach(), gt, lens, s, 0.15), pp.grad) < 1e-5


@pytest.mark.parametrize('B,N,lens', [(5, 300, [300, 253, 252, 127, 1]), (3, 126, [126, 125, 60]), (4, 127, [127, 126, 3, 64]),
                                      (2, 1000, [1000, 881]), (3, 379, [379, 378, 377])])
@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_ff_pair_fused_matches_two_launches_and_torch(ops, precision, B, N, lens):
    """csrc/dx_ffpair.hip (one launch, hidden tensor consumed from LDS, 126-token tiles) against the two-launch path it replaces
    and against fp32 PyTorch on the same bf16-rounded operands: forward pair and input-gradient pair, lengths on both sides of
    the 126-token tile edges."""
    D, Fc = 128, 1024
    H16 = torch.bfloat16 if precision == "bf16" else torch.float16
    ops.set_precision(precision)
    try:
        w1 = randn(Fc, D, 3, seed=1, scale=1.0 / math.sqrt(3 * D))
        b1 = randn(Fc, seed=2, scale=0.1)
        w2 = randn(D, Fc, 3, seed=3, scale=1.0 / math.sqrt(3 * Fc))
        b2 = randn(D, seed=4, scale=0.1)
        p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
        L = lens_tensor(lens)
        valid = (torch.arange(N, device=DEV)[None, :] < L[:, None]).float()[:, :, None]
        x = (randn(B, N, D, seed=5) * valid).to(H16)             # like a masked LayerNorm output
        # ---- forward ----
        h_ref = ops.conv_gemm(x, p1, b1, relu=True, lens=L, halo=1, out_dtype=H16)
        z_ref = ops.conv_gemm(h_ref, p2, b2, lens=L, halo=0)
        z, h = ops.ff_pair(x, p1, p2, b1, b2, L)
        xf, w1f, w2f = x.float(), w1.to(H16).float(), w2.to(H16).float()
        h_t = torch.relu(F.conv1d(xf.transpose(1, 2), w1f, b1, padding=1)).to(H16).float()
        z_t = F.conv1d(h_t, w2f, b2, padding=1).transpose(1, 2)
        for b, n in enumerate(lens):
            hn = min(n + 1, N)                                               # hidden rows that anything reads: 0 .. len (incl. the halo row)
            assert (h[b, :hn].float() - h_ref[b, :hn].float()).abs().max() <= 2e-2 * h_ref[b, :hn].float().abs().max()
            assert rel_err(z[b, :n], z_ref[b, :n]) < 2e-3
            assert rel_err(z[b, :n], z_t[b, :n]) < 1e-2
            assert rel_err(h[b, :hn].float(), h_t[b, :, :hn].t()) < 1e-2
        assert torch.isfinite(h.float()).all() and torch.isfinite(z).all()   # padding tiles are defined (zero-filled)
        # ---- input-gradient pair: dz -> (mask by h > 0) -> dx, accumulated into the residual-branch gradient ----
        dz = (randn(B, N, D, seed=6) * valid).to(H16)
        base = randn(B, N, D, seed=7)
        dh_ref = ops.conv_gemm(dz, p2, None, transpose=True, relu_aux=h_ref, lens=L, halo=1, out_dtype=H16)
        dx_ref = ops.conv_gemm(dh_ref, p1, None, transpose=True, out=base.clone(), accumulate=True, lens=L, halo=0)
        dx, dh = ops.ff_pair(dz, p1, p2, None, None, L, backward=True, aux=h_ref, out=base.clone(), accumulate=True)
        dh_t = F.conv_transpose1d(dz.float().transpose(1, 2), w2f, padding=1) * (h_t > 0)
        dx_t = base + F.conv_transpose1d(dh_t.to(H16).float(), w1f, padding=1).transpose(1, 2)
        for b, n in enumerate(lens):
            hn = min(n + 1, N)
            assert (dh[b, :hn].float() - dh_ref[b, :hn].float()).abs().max() <= 2e-2 * dh_ref[b, :hn].float().abs().max().clamp_min(1e-6)
            assert rel_err(dx[b, :n], dx_ref[b, :n]) < 2e-3
            assert rel_err(dx[b, :n], dx_t[b, :n]) < 1e-2
        assert torch.isfinite(dh.float()).all() and torch.isfinite(dx).all()
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_fused_pitch_predictor_chain_matches_torch_and_the_layer_launches(ops, precision):
    """csrc/dx_pitch.hip (layers/pitch_predictor.py:38-74 in one launch per direction) against fp32 torch on the folded weights, forward and
    input gradient, at utterance lengths around the 122-token tile (1, a tile, a tile + 1, two tiles - 1, the tensor's end).  The bar is the
    layer-by-layer launches' own distance from torch on the same inputs (16-bit activations flip the ReLU of near-zero pre-activations: ~6 % of
    the gradient norm in bf16 on this random network, for both forms); the ReLU sign bits handed to the backward are checked exactly wherever
    torch's pre-activation is clearly away from zero."""
    from ubisoft_laforge_daft_exprt_amd import loss as L
    ops.set_precision(precision)
    try:
        g = torch.Generator().manual_seed(7)
        state = {}
        for k, shape in L.pitch_predictor_shapes().items():
            if k.endswith('num_batches_tracked'):
                state[k] = torch.tensor(10)
            elif k.endswith('running_var') or k.endswith('weight_g'):
                state[k] = torch.rand(shape, generator=g) + 0.5
            else:
                state[k] = torch.randn(shape, generator=g) * (0.3 if 'weight_v' in k else 0.2)
        layers = L.fold_pitch_predictor(state, DEV, ops.DEFAULT)
        lens_h = [1, 122, 123, 243, 300, 57]
        B, M, T = len(lens_h), 80, 300
        lens = lens_tensor(lens_h)
        valid = (torch.arange(T, device=DEV)[None, :] < lens[:, None])
        vm = valid[:, None, :]
        mel = (randn(B, M, T, seed=3) * vm).contiguous()
        dpp = (randn(B, T, seed=5) * valid).contiguous()
        assert ops.pitch_chain_applies(layers, mel, precision)
        pp, masks = ops.pitch_chain_fwd(mel, layers, lens, precision)
        dmel = torch.zeros_like(mel)
        ops.pitch_chain_bwd(dpp, masks, layers, lens, precision, dmel)
        # torch, fp32, on the same folded weights
        melr = mel.clone().requires_grad_(True)
        x, pre = melr, []
        for l in layers[:3]:
            v = F.conv1d(x, l['w'], l['b'], padding=1)
            pre.append(v.detach())
            x = v.relu() * l['scale'][None, :, None] + l['shift'][None, :, None]
        pp_ref = F.conv1d(x, layers[3]['w'][:1], layers[3]['b'][:1], padding=1)[:, 0]
        (pp_ref * dpp).sum().backward()
        # the layer-by-layer launches (loss.py's other branch)
        hd = ops.hidden_dtype(precision)
        xl, acts = ops.transpose(mel), []
        for i, layer in enumerate(layers[:-1]):
            r = ops.conv_gemm(xl, layer['pack'], layer['b'], relu=True, lens=lens, halo=3 - i, prec=precision, out_dtype=hd)
            acts.append(r)
            xl = ops.channel_affine(r, layer['scale'], layer['shift'], prec=precision)
        pp_l = ops.conv_gemm(xl, layers[-1]['pack'], layers[-1]['b'], lens=lens, halo=0, prec=precision)[..., 0]
        gl = torch.zeros(B, T, 4, device=DEV)
        gl[..., 0] = dpp
        for k in range(3, 0, -1):
            prev = layers[k - 1]
            gl = ops.conv_gemm(gl, layers[k]['pack'], None, transpose=True, post_scale=prev['scale'], post_shift=prev['zeros'], relu_aux=acts[k - 1],
                               lens=lens, halo=4 - k, prec=precision, out_dtype=hd)
        dmel_l = ops.transpose(ops.conv_gemm(gl, layers[0]['pack'], None, transpose=True, lens=lens, halo=0, prec=precision))

        def dist(a, b, m):
            return float(((a - b) * m).norm() / (b * m).norm())
        e_pp, e_pp_l = dist(pp, pp_ref.detach(), valid), dist(pp_l, pp_ref.detach(), valid)
        e_g, e_g_l = dist(dmel, melr.grad, vm), dist(dmel_l, melr.grad, vm)
        bar_pp, bar_g = {'bf16': (3e-2, 9e-2), 'fp16': (4e-3, 4e-2)}[precision]      # (measured: bf16 2.1e-2 / 6.3e-2, fp16 2.5e-2 for the gradient; the layer launches 2.2e-2 / 6.5e-2 / 2.6e-2)
        assert e_pp < bar_pp and e_pp < 1.25 * e_pp_l + 1e-3, (e_pp, e_pp_l)
        assert e_g < bar_g and e_g < 1.25 * e_g_l + 1e-3, (e_g, e_g_l)
        assert float((dmel * ~vm).abs().max()) == 0.0                                     # nothing written beyond the lengths
        for l in range(3):
            bits = masks[:, :, l].contiguous()                                            # (B, T, 8) int32
            sign = ((bits[:, :, :, None] >> torch.arange(32, device=DEV)) & 1).reshape(B, T, 256).bool()
            v0 = pre[l].transpose(1, 2)                                                   # (B, T, 256)
            clear = (v0.abs() > 0.05) & valid[:, :, None]
            assert bool(((sign == (v0 > 0)) | ~clear).all()), l
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
@pytest.mark.parametrize('lens_h,N,p', [([257, 64, 1, 200], 257, 0.1), ([150, 149, 7, 128, 65, 33, 150, 2, 90], 150, 0.0), ([83, 120], 120, 0.1)])
def test_attention_projection_layernorm_in_one_launch_equals_the_two_launches_bitwise(ops, precision, lens_h, N, p):
    """dx_attention_proj_ln_fwd == dx_attention_fwd + dx_proj_ln_fwd, bit for bit (context, lse, z, y, its 16-bit copy, mean, rstd), with
    dropout in both places, ragged lengths (a 1-token utterance, lengths at and around the 64-query tile), padding tiles, 9 utterances
    (the utterance -> XCD numbering pads to a multiple of 8)."""
    ops.set_precision(precision)
    try:
        h16 = {'bf16': torch.bfloat16, 'fp16': torch.float16}[precision]
        B = len(lens_h)
        lens = lens_tensor(lens_h)
        qkv = randn(B, N, 384, seed=1).to(h16)
        res = randn(B, N, 128, seed=2)
        wout, bout = randn(128, 128, seed=3, scale=0.09), randn(128, seed=4, scale=0.1)
        ln_w, ln_b = 1 + randn(128, seed=5, scale=0.1), randn(128, seed=6, scale=0.1)
        pack = ops.PackedWeight(wout)
        so = torch.tensor([5], dtype=torch.int64, device=DEV)
        ctx0, lse0 = ops.attention_fwd(qkv, lens, 2, 11, p, ctx_dtype=h16, seed_offset=so)
        z0, y0, m0, r0, yh0 = ops.proj_ln_fwd(ctx0, pack, bout, res, ln_w, ln_b, None, lens, seed_pre=12, p_pre=p, shadow=True, seed_offset=so)
        assert ops.attn_proj_ln_applies(qkv, 2, pack, precision)
        ctx1, lse1, z1, y1, m1, r1, yh1 = ops.attn_proj_ln_fwd(qkv, lens, 2, 11, p, pack, bout, res, ln_w, ln_b, None, seed_pre=12, p_pre=p, shadow=True,
                                                               seed_offset=so)
        for name, a, b in (('ctx', ctx0, ctx1), ('lse', lse0, lse1), ('z', z0, z1), ('y', y0, y1), ('y16', yh0, yh1), ('mean', m0, m1), ('rstd', r0, r1)):
            assert torch.equal(a, b), name
    finally:
        ops.set_precision('f32')
