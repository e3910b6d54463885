import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd.utils import knn
B, C, N, k = 1, 64, 1024, 20
rng = np.random.default_rng(1)
x = rng.standard_normal((B, C, N)).astype(np.float32)
idx = knn(torch.from_numpy(x).cuda(), k).cpu().numpy()[0]
xd = x.astype(np.float64)[0]
sq = (xd ** 2).sum(0)
d = sq[:, None] + sq[None, :] - 2 * xd.T @ xd
true = np.argsort(d, axis=1, kind="stable")[:, :k]
miss = [sorted(set(true[q]) - set(idx[q])) for q in range(N)]
allm = np.array([m for q in range(N) for m in miss[q]])
print("missing total", len(allm), "per query mean", len(allm) / N)
print("missing idx mod 128 histogram (by 8):", np.bincount((allm % 128) // 8, minlength=16))
print("missing idx // 128:", np.bincount(allm // 128, minlength=8))
print("missing idx mod 8:", np.bincount(allm % 8, minlength=8))
qm = np.array([q for q in range(N) for m in miss[q]])
print("query mod 128 //8:", np.bincount((qm % 128) // 8, minlength=16))
print("rank of missing in true order:", np.bincount([list(true[q]).index(m) for q in range(N) for m in miss[q]], minlength=k))
for q in (0, 1, 500):
    print("q", q, "returned", idx[q].tolist())
    print("   true   ", true[q].tolist())
    print("   d ret ", np.round(d[q, idx[q]], 2).tolist())
    print("   d true", np.round(d[q, true[q]], 2).tolist())
