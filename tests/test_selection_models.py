"""CPU models of three round-3 kernels' ALGORITHMS (no GPU, no library call): what the HIP code relies on is checked here in numpy /
torch-fp64, the kernels themselves in tests/test_gpu_parity.py, test_gpu_ref_kernels.py and test_gpu_grad_routes.py.

* knn_select.hip (pointnet2 `knn_kernel_fast`, utils/lib/src/interpolate_gpu.cu:9-57, for large k or many candidates): the bound from
  256 bucket minima is valid for every cloud (>= k candidates reach it), its bisection ends inside [k, k + 8] buckets or at an exact
  value, the survivors' rank count is the reference's order (ascending squared distance, lowest index first), and the exact
  two-bisection path leaves exactly k survivors under heavy ties.
* knn_small.hip (three_nn / k <= 4, interpolate_gpu.cu:81-124): a lane's 32-candidate hit mask is built against the k-th best at the TOP of
  the block (stale inside it), hits are popped lowest index first and inserted with a strict '<' into four sorted slots -- the result is
  still the stable order, and slots beyond the candidate count keep (+inf, 0).
* l3d_layernorm_ref_backward (utils/transformer.py:109-119: unbiased std, eps added to std): the closed form the kernel evaluates
  equals autograd of the reference's op sequence."""
import numpy as np
import torch


def _dist_bits(q, c):
    """squared distances of one query to all candidates, the reference's rounding sequence, as ordered uint32 keys"""
    d = (c - q).astype(np.float32)
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    return d2, d2.view(np.uint32)


def _bound(bits, k, slack=8):
    """knn_select_kernel's T0: bisection over the bit pattern on counts of BUCKET MINIMA (bucket = index mod 256)"""
    m = len(bits)
    pad = np.full((-m) % 256, 0x7F800000, np.uint32)                      # padding candidates: +inf
    bm = np.concatenate([bits, pad]).reshape(-1, 256).min(axis=0)
    lo, hi, steps = 0, 0x7FFFFFFF, 0
    while lo < hi:
        mid = lo + ((hi - lo) >> 1)
        c = int((bm <= mid).sum())
        steps += 1
        if c >= k:
            hi = mid
            if c <= k + slack:
                break
        else:
            lo = mid + 1
    return np.uint32(hi), bm, steps


def _select_model(q, c, k, cap=512):
    d2, bits = _dist_bits(q, c)
    T0, bm, steps = _bound(bits, k)
    assert int((bm <= T0).sum()) >= k                                      # >= k buckets, i.e. >= k DISTINCT candidates, reach T0
    surv = np.nonzero(bits <= T0)[0]
    assert len(surv) >= k
    exact = len(surv) > cap
    if exact:                                                              # the kernel's exact path: k-th key by two bisections
        lo, hi = 0, 0xFFFFFFFF
        while lo < hi:
            mid = lo + ((hi - lo) >> 1)
            if int((bits <= mid).sum()) >= k:
                hi = mid
            else:
                lo = mid + 1
        T = np.uint32(hi)
        need = k - int((bits < T).sum())
        assert need >= 1
        ilo, ihi = 0, len(bits) - 1
        ties = bits == T
        while ilo < ihi:
            mid = ilo + ((ihi - ilo) >> 1)
            if int((ties & (np.arange(len(bits)) <= mid)).sum()) >= need:
                ihi = mid
            else:
                ilo = mid + 1
        surv = np.nonzero((bits < T) | (ties & (np.arange(len(bits)) <= ihi)))[0]
        assert len(surv) == k
    keys = (bits[surv].astype(np.uint64) << np.uint64(32)) | surv.astype(np.uint64)
    rank = (keys[None, :] < keys[:, None]).sum(axis=1)                     # rank by counting: keys are distinct
    out = np.empty(k, np.int64)
    sel = rank < k
    out[rank[sel]] = surv[sel]
    return out, d2, len(surv), exact, steps


def test_knn_select_algorithm_model():
    rng = np.random.default_rng(5)
    seen_exact = False
    for (m, k, kind) in [(8192, 64, "uniform"), (8192, 64, "sorted"), (5000, 200, "normal"), (1024, 16, "uniform"), (300, 33, "uniform"),
                         (4097, 100, "grid"), (2048, 64, "same"), (8192, 1, "uniform"), (8192, 64, "clustered")]:
        if kind == "uniform":
            c = rng.uniform(-1, 1, (m, 3))
        elif kind == "normal":
            c = np.clip(rng.standard_normal((m, 3)), -2, 2)
        elif kind == "sorted":
            c = rng.uniform(-1, 1, (m, 3)); c = c[np.argsort(c[:, 0])]
        elif kind == "grid":
            c = rng.integers(0, 4, (m, 3)) / 3.0
        elif kind == "clustered":
            c = rng.standard_normal((m, 3)) * 0.01 + rng.standard_normal((1, 3))
        else:
            c = np.full((m, 3), 0.25)
        c = c.astype(np.float32)
        for qi in range(6):
            q = c[rng.integers(0, m)] if qi % 2 else rng.uniform(-1, 1, 3).astype(np.float32)
            got, d2, nsurv, exact, steps = _select_model(q, c, k)
            want = np.argsort(d2, kind="stable")[:k]                       # ascending distance, lowest index first on ties
            assert np.array_equal(got, want), (m, k, kind)
            assert steps <= 31
            seen_exact |= exact
            if kind in ("uniform", "normal", "sorted", "clustered") and k >= 16 and m >= 1024:
                # the bound is tight on clouds without massive ties, whatever their ORDER (buckets stride the index space):
                # about k (256 / (256 - k)) ... a few times k survivors, never the exact path
                assert not exact and nsurv <= 4 * k + 64, (m, k, kind, nsurv)
    assert seen_exact                                                      # the grid / single-point clouds went through the exact path


def test_layernorm_backward_closed_form():
    torch.manual_seed(3)
    for C in (12, 64, 512):
        x = (torch.randn(7, C, dtype=torch.float64) * 2 + 0.5).requires_grad_()
        a = (torch.randn(C, dtype=torch.float64) * 0.5 + 1).requires_grad_()
        b = torch.randn(C, dtype=torch.float64).requires_grad_()
        g = torch.randn(7, C, dtype=torch.float64)
        eps = 1e-6
        y = a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + eps) + b       # the reference's op sequence
        (y * g).sum().backward()
        with torch.no_grad():                                              # what layernorm_ref_backward_kernel evaluates
            xc = x - x.mean(-1, keepdim=True)
            s = torch.sqrt((xc * xc).sum(-1, keepdim=True) / (C - 1))
            d = s + eps
            dz = g * a
            t1 = (dz * xc).sum(-1, keepdim=True)
            dxc = dz / d - (t1 / (d * d * (C - 1) * s)) * xc
            dx = dxc - dxc.mean(-1, keepdim=True)
            da = (g * xc / d).sum(0)
            db = g.sum(0)
        assert torch.allclose(dx, x.grad, rtol=1e-10, atol=1e-12)
        assert torch.allclose(da, a.grad, rtol=1e-10, atol=1e-12)
        assert torch.allclose(db, b.grad, rtol=1e-10, atol=1e-12)


def _small_model(q, c, k):
    """one lane of knn_small_kernel"""
    d2, bits = _dist_bits(q, c)
    bd = [np.float32(np.inf)] * 4
    bi = [0] * 4
    thr = np.uint32(0x7F800000)
    pops = 0
    for g0 in range(0, len(c), 32):
        blk = range(g0, min(g0 + 32, len(c)))
        hits = [j for j in blk if bits[j] < thr]                           # the mask: against the threshold at the top of the block
        for j in hits:                                                     # popped highest bit = lowest index first
            pops += 1
            d = d2[j]
            c3, c2, c1, c0 = d < bd[3], d < bd[2], d < bd[1], d < bd[0]
            bd[3], bi[3] = (bd[2], bi[2]) if c2 else ((d, j) if c3 else (bd[3], bi[3]))
            bd[2], bi[2] = (bd[1], bi[1]) if c1 else ((d, j) if c2 else (bd[2], bi[2]))
            bd[1], bi[1] = (bd[0], bi[0]) if c0 else ((d, j) if c1 else (bd[1], bi[1]))
            bd[0], bi[0] = (d, j) if c0 else (bd[0], bi[0])
        if hits:
            thr = np.float32(bd[k - 1]).view(np.uint32)
    return np.array(bi[:k]), np.array(bd[:k], np.float32), d2, pops


def test_knn_small_algorithm_model():
    rng = np.random.default_rng(9)
    for (m, k, kind) in [(1024, 3, "uniform"), (1000, 4, "grid"), (300, 1, "uniform"), (70, 2, "same"), (2, 3, "uniform"), (2049, 3, "sorted")]:
        if kind == "uniform":
            c = rng.uniform(-1, 1, (m, 3))
        elif kind == "sorted":
            c = rng.uniform(-1, 1, (m, 3)); c = c[np.argsort(c[:, 0])]
        elif kind == "grid":
            c = rng.integers(0, 4, (m, 3)) / 3.0
        else:
            c = np.full((m, 3), 0.25)
        c = c.astype(np.float32)
        for qi in range(5):
            q = c[rng.integers(0, m)] if qi % 2 else rng.uniform(-1, 1, 3).astype(np.float32)
            idx, val, d2, pops = _small_model(q, c, k)
            order = np.argsort(d2, kind="stable")[:k]
            n = min(k, m)
            assert np.array_equal(idx[:n], order[:n]) and np.array_equal(val[:n], d2[order[:n]]), (m, k, kind)
            assert all(np.isinf(val[n:])) and all(idx[n:] == 0)            # slots beyond the candidate count: (+inf, 0)
            if kind == "uniform" and m >= 1024:
                assert pops <= 32 + 12 * k * int(np.log2(m / 32) + 1)      # the stale threshold costs a few extra pops, not a rescan
