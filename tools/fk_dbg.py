import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd._lib import lib, ptr, stream_ptr, check
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, N, k = 1, 1024, 20
x = torch.from_numpy(np.random.default_rng(1).standard_normal((B, C, N)).astype(np.float32)).cuda()
nb = lib().l3d_knn_feature_workspace_bytes(B, C, N)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
idx = torch.zeros((B, N, k), dtype=torch.int64, device="cuda")
check(lib().l3d_knn_feature(ptr(x), B, C, N, k, ptr(ws), ptr(idx), stream_ptr()), "knn")
torch.cuda.synchronize()
Cp = (C + 63) // 64 * 64
off = B * Cp * N * 4 + B * N * 8
dbg = ws[off:off + B * N * 32 * 4].view(torch.float32).view(N, 32).cpu().numpy()
xd = x.cpu().numpy().astype(np.float64)[0]; sq = (xd ** 2).sum(0); d = sq[:, None] + sq[None, :] - 2 * xd.T @ xd
kth = -np.sort(d, axis=1)[:, 19]
for q in (0, 1, 500):
    print("q", q, "M", dbg[q, 31], "true k-th p0", kth[q] + sq[q])
    for l in range(4):
        print("    lane", l, "thr %.3f margin %.3f top0 %.3f topT %.3f offer %.3f cnt0 %.0f cntT %.0f mr00 %.3f" % tuple(dbg[q, l * 8:l * 8 + 8]))
print("idx[1]", idx[0, 1].tolist())
