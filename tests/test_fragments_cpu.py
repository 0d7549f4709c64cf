"""CPU: the torch restatement of the FRAGMENT slot format (include/inerf.h, csrc/layout.h) that the GPU tests hold the kernels
against - encode / decode round trips for every slot width, the layout formula of the header spelled out, and the
per-point normalisers of the gradient slots."""
import pytest
import torch

from intrinsicnerf_amd import kernels


@pytest.mark.parametrize("width", [256, 128, 64, 32])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 200])
def test_fragment_round_trip(width, n):
    g = torch.Generator().manual_seed(width + n)
    x = torch.randn(n, width, generator=g) * torch.logspace(-3, 1, width)[None, :]
    frag = kernels.frag_encode(x)
    tiles = (n + 63) // 64
    assert frag.dtype == torch.float16 and frag.numel() == tiles * 64 * width * 2          # 4 bytes per element, whole tiles
    back = kernels.frag_decode(frag, n, width=width)
    assert back.shape == (n, width)
    assert float((back - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())              # hi + lo carry 22 bits
    # padding points of the last tile are zeros
    full = kernels.frag_decode(frag, tiles * 64, width=width)
    assert float(full[n:].abs().max()) == 0.0 if tiles * 64 > n else True


@pytest.mark.parametrize("width", [256, 64])
def test_fragment_layout_is_the_headers_formula(width):
    """byte offset = (((tile * 4 + kb) * (width / 32) + cb) * 2 + plane) * 1024 + lane * 16 + 2 * i;  channel = 32 cb + (lane & 31),
    point = 64 tile + 32 (kb >> 1) + (i & 3) + 8 ((i >> 2) + 2 (kb & 1)) + 4 (lane >> 5)   (include/inerf.h)."""
    n = 128
    x = (torch.arange(n)[:, None] * 256 + torch.arange(width)[None, :]).float() / 8.0       # 8 x = 15-bit integers: exact as f16 hi + lo
    frag = kernels.frag_encode(x)
    cbs = width // 32
    g = torch.Generator().manual_seed(0)
    for _ in range(200):
        tile, kb, cb, lane, i = (int(torch.randint(0, m, (1,), generator=g)) for m in (2, 4, cbs, 64, 8))
        half0 = ((((tile * 4 + kb) * cbs + cb) * 2 + 0) * 1024 + lane * 16 + 2 * i) // 2
        half1 = ((((tile * 4 + kb) * cbs + cb) * 2 + 1) * 1024 + lane * 16 + 2 * i) // 2
        chan = 32 * cb + (lane & 31)
        point = 64 * tile + 32 * (kb >> 1) + (i & 3) + 8 * ((i >> 2) + 2 * (kb & 1)) + 4 * (lane >> 5)
        assert float(frag[half0]) + float(frag[half1]) == float(x[point, chan]) * 8.0


def test_gradient_fragments_carry_per_point_normalisers():
    g = torch.Generator().manual_seed(1)
    n = 150
    x = torch.randn(n, 256, generator=g) * torch.logspace(0, -12, n)[:, None]              # thirteen decades between the points
    x[7] = 0.0                                                                              # a point without gradient
    frag, scales = kernels.grad_frag_encode(x)
    assert scales.numel() == 192 + 64 and float(scales[7]) == 1.0 and float(scales[n:].min()) == 1.0
    m = x.abs().amax(1)
    ok = m > 0
    assert bool(((scales[:n][ok] > m[ok]) & (scales[:n][ok] <= 2 * m[ok])).all())          # the power of two above the point's maximum
    back = kernels.grad_frag_decode(frag, scales, n)
    assert float(((back - x).abs() / m.clamp_min(1e-30)[:, None]).max()) <= 2.0 ** -21    # 22 bits relative to the POINT's scale
