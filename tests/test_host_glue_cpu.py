"""Host-side autograd glue that needs no GPU."""
import torch

from ubisoft_laforge_daft_exprt_amd import functional as Fx


def test_split_film_matches_per_block_slicing():
    """SplitFilmFn (one copy forward, one stack backward) == slicing film[:, i, :] per FFT block, values and gradients,
    including blocks whose slice receives no gradient."""
    g = torch.Generator().manual_seed(0)
    film = torch.randn(5, 4, 256, generator=g, requires_grad=True)
    ref = film.detach().clone().requires_grad_(True)
    parts = Fx.SplitFilmFn.apply(film)
    assert len(parts) == 4 and all(p.is_contiguous() and p.shape == (5, 256) for p in parts)
    ws = [torch.randn(5, 256, generator=g) for _ in range(4)]
    used = (0, 1, 3)                                       # block 2 contributes nothing
    sum((parts[i] * ws[i]).sum() for i in used).backward()
    sum((ref[:, i, :] * ws[i]).sum() for i in used).backward()
    for i in range(4):
        assert torch.equal(parts[i], ref[:, i, :])
    assert torch.equal(film.grad, ref.grad)
    assert torch.count_nonzero(film.grad[:, 2]) == 0


def test_exchange_groups_leave_at_most_a_quarter_of_the_gradient_bytes_exposed():
    """trainer.group_of over the reference's 184 parameters (reversed registration order, as ddp.GradientReducer packs them): every
    group is ONE contiguous run (so groups never share a bucket and need no tiny extra all-reduce), the groups appear in phase order,
    and the group launched after the last backward kernel holds <= 25 % of the 57.5 MB."""
    from tests import helpers
    from ubisoft_laforge_daft_exprt_amd.trainer import group_of
    shapes = helpers.manifest()['model']
    names = list(shapes)[::-1]
    numel = lambda k: int(__import__('math').prod(shapes[k])) if shapes[k] else 1
    total = sum(numel(k) for k in names)
    for levels in (0, 1, 2, 3):
        gids = [group_of(k, levels) for k in names]
        runs = [g for i, g in enumerate(gids) if i == 0 or g != gids[i - 1]]
        assert runs == list(range(levels + 1)), (levels, runs)
        last = sum(numel(k) for k, g in zip(names, gids) if g == levels)
        if levels == 3:
            assert last / total <= 0.25 and last / total < 0.03, last / total     # prenet layer 0 + prosody embeddings: ~1 MB
        if levels == 2:
            assert 0.25 < last / total < 0.28                                   # the whole prenet: why there is a third cut


def test_dropout_counter_hash_statistics():
    """csrc/dx_common.h ``dx_rand64`` restated in numpy (uint32 arithmetic): four 16-bit fields per 64-bit draw decide keep / drop
    by ``field >= round(p * 65536)``.  On attention-shaped counters ((row << 14) | key group) and on consecutive counters: keep rate
    per field, cross-field, lag-1 along keys / rows, seed vs seed + 1 correlation at noise level, top-byte uniformity, and the
    variance of the number of dropped elements per row against the binomial's."""
    import numpy as np
    M = np.uint32

    def rand64(seed, idx):
        with np.errstate(over='ignore'):
            x = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ M(seed & 0xFFFFFFFF)
            hi = (idx >> np.uint64(32)).astype(np.uint32) ^ M(seed >> 32)
            x ^= hi * M(0x9E3779B1)
            x ^= x >> M(16); x = x * M(0x7FEB352D); x ^= x >> M(15); x = x * M(0x846CA68B); x ^= x >> M(16)
            y = (x ^ M(0x85EBCA6B)) * M(0xC2B2AE35); y ^= y >> M(15)
        return [x & M(0xFFFF), x >> M(16), y & M(0xFFFF), y >> M(16)]

    thr = int(round(0.1 * 65536))
    corr = lambda a, b: abs(float(np.corrcoef(a.ravel(), b.ravel())[0, 1]))
    for seed in (0x1234567890ABCDEF, 7):
        rows = np.arange(2048, dtype=np.uint64)[:, None]
        kg = np.arange(224, dtype=np.uint64)[None, :]
        F = rand64(seed, (rows << np.uint64(14)) | kg)
        keep = [(f >= thr).astype(np.float64) for f in F]
        n = keep[0].size
        noise = 1.0 / np.sqrt(n)
        assert all(abs(k.mean() - 0.9) < 4 * 0.3 * noise for k in keep), [k.mean() for k in keep]
        assert max(corr(keep[i], keep[j]) for i in range(4) for j in range(i + 1, 4)) < 5 * noise
        assert max(corr(k[:, 1:], k[:, :-1]) for k in keep) < 5 * noise          # neighbouring key groups
        assert max(corr(k[1:, :], k[:-1, :]) for k in keep) < 5 * noise          # neighbouring query rows
        other = [(f >= thr).astype(np.float64) for f in rand64(seed + 1, (rows << np.uint64(14)) | kg)]
        assert max(corr(a, b) for a, b in zip(keep, other)) < 5 * noise          # the next seed is an unrelated stream
        for f in F:                                                              # top byte of every field: chi-square, 255 dof
            h = np.bincount((f >> M(8)).ravel().astype(np.int64), minlength=256)
            assert ((h - n / 256) ** 2 / (n / 256)).sum() < 255 + 6 * np.sqrt(2 * 255)
        dropped = sum(1 - k for k in keep).sum(axis=1)
        assert 0.85 < dropped.var() / (4 * 224 * 0.1 * 0.9) < 1.15
        seq = [(f >= thr).astype(np.float64) for f in rand64(seed, np.arange(1 << 20, dtype=np.uint64))]
        assert max(corr(k[1:], k[:-1]) for k in seq) < 5e-3 and abs(np.mean([k.mean() for k in seq]) - 0.9) < 2e-3
