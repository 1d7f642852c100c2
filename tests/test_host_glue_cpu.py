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
