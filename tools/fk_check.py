"""featknn correctness sweep: every returned neighbour against exact fp64 distances (the parity test's criterion), per shape"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd.utils import knn
for (B, C, N, k) in [(2, 64, 300, 16), (2, 64, 1024, 20), (2, 128, 1024, 20), (2, 128, 300, 20), (1, 256, 513, 7), (2, 96, 256, 20), (1, 130, 257, 64), (1, 5, 40, 33), (2, 192, 1024, 20)]:
    rng = np.random.default_rng(C + N)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = knn(torch.from_numpy(x).cuda(), k).cpu().numpy()
    xd = x.astype(np.float64)
    sq = (xd ** 2).sum(axis=1)
    d = sq[:, :, None] + sq[:, None, :] - 2 * np.einsum("bci,bcj->bij", xd, xd)
    kth = np.sort(d, axis=-1)[:, :, k - 1]
    ok_range = (idx >= 0).all() and (idx < N).all()
    got = np.take_along_axis(d, np.clip(idx, 0, N - 1), axis=-1)
    tol = 4e-6 * sq.max()
    bad = (got.max(axis=-1) > kth + tol).sum()
    unsorted = (np.diff(got, axis=-1) < -tol).sum()
    print(f"B{B} C{C} N{N} k{k}: in range {ok_range}  queries with a too-far neighbour {bad}  order violations {unsorted}  self-first {(idx[:, :, 0] == np.arange(N)[None]).mean():.4f}", flush=True)
