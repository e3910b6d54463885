"""Feature-space kNN (l3d_knn_feature) against exact fp64 distances + timing vs the torch op sequence."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learning3d_amd.utils import knn

def ref_ops(xf, k):
    inner = -2 * torch.matmul(xf.transpose(2, 1), xf)
    xx = torch.sum(xf ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]

def check(B, C, N, k, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = knn(torch.from_numpy(x).cuda(), k).cpu().numpy()
    xd = x.astype(np.float64)
    g = np.einsum("bci,bcj->bij", xd, xd)
    sq = (xd ** 2).sum(1)
    d = sq[:, :, None] + sq[:, None, :] - 2 * g
    kth = np.sort(d, axis=-1)[:, :, k - 1]
    got = np.take_along_axis(d, idx, axis=-1)
    scale = sq.max()
    ok_set = np.all(got.max(-1) <= kth + 2e-6 * scale)
    ok_sorted = np.all(np.diff(got, axis=-1) >= -4e-6 * scale)
    ok_unique = all(len(set(r)) == k for r in idx.reshape(-1, k)[:: max(1, B * N // 500)])
    ok_self = np.all(idx[:, :, 0] == np.arange(N)[None])
    print(f"B={B} C={C} N={N} k={k}: set {ok_set} sorted {ok_sorted} unique {ok_unique} self-first {ok_self}", flush=True)
    return ok_set and ok_sorted and ok_unique and ok_self

ok = True
for cfg in [(2, 64, 300, 16, 1), (2, 32, 128, 20, 2), (3, 128, 1000, 20, 3), (1, 256, 513, 7, 4), (2, 64, 1024, 20, 5), (1, 96, 77, 1, 6)]:
    ok &= check(*cfg)
print("ALL OK" if ok else "FAILED")
for (B, C, N, k) in [(32, 64, 1024, 20), (32, 128, 1024, 20), (32, 256, 1024, 20), (8, 64, 4096, 20)]:
    x = torch.randn(B, C, N, device="cuda")
    for name, fn in (("hip", lambda: knn(x, k)), ("torch ops", lambda: ref_ops(x, k))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"  B={B} C={C} N={N} k={k} {name:10s} {dt*1e6:9.1f} us")
