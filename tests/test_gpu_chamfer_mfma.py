"""Chamfer search with the candidates ranked on the fp16 matrix cores (chamfer_mfma.hip, l3d_chamfer_forward_variant 3; the
default from 2^24 pairs per cloud) against the exact per-pair kernel (variant 2 = chamfer_fwd_packed_kernel, itself pinned bit
for bit to the reference's C++ and CUDA kernels in test_gpu_parity.py / test_gpu_ref_kernels.py): distances AND indices must be
identical -- the matrix cores only choose which candidates get the exact evaluation.  Inputs are picked to stress the error
band: exact ties (lattices, duplicated points), clouds far from the origin, tiny and huge extents, ragged sizes, one-point
clouds, non-finite coordinates (which take the kernel's exact-scan path)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run(a, b, variant):
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    B, N, _ = a.shape
    M = b.shape[1]
    d1, d2 = torch.empty((B, N), device="cuda"), torch.empty((B, M), device="cuda")
    i1 = torch.empty((B, N), dtype=torch.int32, device="cuda")
    i2 = torch.empty((B, M), dtype=torch.int32, device="cuda")
    check(lib().l3d_chamfer_forward_variant(ptr(a), ptr(b), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), variant, stream_ptr()), "chamfer")
    torch.cuda.synchronize()
    return d1, d2, i1, i2


def same(a, b):
    got, want = run(a, b, 3), run(a, b, 2)
    for g, w, name in zip(got, want, ("dist1", "dist2", "idx1", "idx2")):
        if g.dtype.is_floating_point:                       # NaN rows (non-finite inputs) compare as bit patterns
            assert torch.equal(g.view(torch.int32), w.view(torch.int32)), name
        else:
            assert torch.equal(g, w), (name, int((g != w).sum()))


def clouds(seed, B, N, M, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        a, b = torch.rand((B, N, 3), generator=g) - 0.5, torch.rand((B, M, 3), generator=g) - 0.5
    elif kind == "offset":                                  # far from the origin: the centring has to carry it
        a, b = torch.rand((B, N, 3), generator=g) + 1000.0, torch.rand((B, M, 3), generator=g) + 1000.0
    elif kind == "tiny":
        a, b = torch.rand((B, N, 3), generator=g) * 1e-6, torch.rand((B, M, 3), generator=g) * 1e-6
    elif kind == "huge":
        a, b = torch.rand((B, N, 3), generator=g) * 1e12, torch.rand((B, M, 3), generator=g) * 1e12 + 3e11
    elif kind == "lattice":                                 # integer lattice: masses of exact ties, lowest index must win
        a = torch.randint(0, 12, (B, N, 3), generator=g).float() * 0.25
        b = torch.randint(0, 12, (B, M, 3), generator=g).float() * 0.25
    elif kind == "duplicates":
        a = torch.rand((B, N, 3), generator=g)
        b = a[:, torch.randint(0, N, (M,), generator=g)].clone()      # every b point is a copy of some a point: distance exactly 0
    elif kind == "disjoint":                                # two separated clusters: the nearest neighbour is far away
        a, b = torch.rand((B, N, 3), generator=g), torch.rand((B, M, 3), generator=g) + 5.0
    elif kind == "gauss":
        a, b = torch.randn((B, N, 3), generator=g), torch.randn((B, M, 3), generator=g) * 3.0
    elif kind == "planar":                                  # one axis constant
        a, b = torch.rand((B, N, 3), generator=g), torch.rand((B, M, 3), generator=g)
        a[..., 2] = 0.125; b[..., 2] = 0.125
    else:
        raise ValueError(kind)
    return a.cuda().contiguous(), b.cuda().contiguous()


@pytest.mark.parametrize("kind", ["uniform", "offset", "tiny", "huge", "lattice", "duplicates", "disjoint", "gauss", "planar"])
@pytest.mark.parametrize("B,N,M", [(2, 1024, 1024), (3, 700, 1900), (1, 33, 5000), (2, 4096, 513)])
def test_mfma_ranked_chamfer_equals_exact_kernel(kind, B, N, M):
    same(*clouds(3, B, N, M, kind))


@pytest.mark.parametrize("kind", ["uniform", "sorted", "patches", "lattice", "duplicates", "offset"])
def test_mfma_ranked_chamfer_sampled_first_pass(kind):
    """From 8192 points per cloud pass 1 only looks at every 4th group of 32 candidates (an upper bound of each query's minimum):
    clouds whose index order is spatial -- sorted along x, or PCN's fine output (16 consecutive points per coarse centre) -- make
    that sample unrepresentative for some queries; the answer must not care."""
    g = torch.Generator().manual_seed(17)
    N, M = 8192, 9000
    if kind in ("sorted", "patches"):
        a, b = torch.rand((1, N, 3), generator=g), torch.rand((1, M, 3), generator=g)
        if kind == "sorted":
            a = a[:, torch.argsort(a[0, :, 0])]
            b = b[:, torch.argsort(b[0, :, 0])]
        else:
            ca, cb = torch.rand((1, N // 16, 1, 3), generator=g), torch.rand((1, M // 8, 1, 3), generator=g)
            a = (ca + 0.01 * torch.rand((1, N // 16, 16, 3), generator=g)).reshape(1, N, 3)
            b = (cb + 0.01 * torch.rand((1, M // 8, 8, 3), generator=g)).reshape(1, M, 3)
        a, b = a.cuda().contiguous(), b.cuda().contiguous()
    else:
        a, b = clouds(19, 1, N, M, kind)
    same(a, b)


def test_mfma_ranked_chamfer_one_point_and_identical_points():
    a, b = clouds(5, 2, 1, 777, "uniform")
    same(a, b)
    same(b, a)
    c = torch.full((2, 600, 3), 0.375, device="cuda")       # zero extent: every candidate ties, index 0 wins everywhere
    same(c, c.clone())
    d1, d2, i1, i2 = run(c, c.clone(), 3)
    assert int(i1.abs().sum()) == 0 and float(d1.abs().sum()) == 0.0


def test_mfma_ranked_chamfer_non_finite_inputs_take_the_exact_scan():
    a, b = clouds(7, 2, 900, 1100, "uniform")
    a[0, 17, 1] = float("nan")
    b[1, 5, 0] = float("inf")
    same(a, b)


def test_mfma_ranked_chamfer_config4_size_one_cloud():
    """BASELINE configs[3]'s per-cloud size, 16384 x 16384, where l3d_chamfer_forward picks this kernel by itself."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    a, b = clouds(11, 2, 16384, 16384, "uniform")
    same(a, b)
    d1, d2 = torch.empty((2, 16384), device="cuda"), torch.empty((2, 16384), device="cuda")
    i1, i2 = torch.empty((2, 16384), dtype=torch.int32, device="cuda"), torch.empty((2, 16384), dtype=torch.int32, device="cuda")
    check(lib().l3d_chamfer_forward(ptr(a), ptr(b), 2, 16384, 16384, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "chamfer")
    w = run(a, b, 2)
    assert torch.equal(d1, w[0]) and torch.equal(i2, w[3])
