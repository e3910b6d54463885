"""kNN at the bench shape: matrix-core kernel (+ fix-up launch) vs the insertion kernel.  usage: knn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd._lib import check, lib, ptr, stream_ptr

def run(x, k, variant, iters=50):
    B, N, _ = x.shape
    idx = torch.empty((B, N, k), dtype=torch.int64, device=x.device)
    f = lambda: check(lib().l3d_knn_graph_variant(ptr(x), B, N, k, ptr(idx), variant, stream_ptr()), "knn")
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, idx

g = torch.Generator().manual_seed(0)
for B, N, k in ((32, 1024, 20), (32, 2048, 20), (32, 512, 16)):
    x = torch.rand((B, N, 3), generator=g).cuda()
    t2, i2 = run(x, k, 2)
    t1, i1 = run(x, k, 1)
    print(f"B={B} N={N} k={k}: mfma {t2:7.1f} us   insertion {t1:7.1f} us   equal={bool((i1 == i2).all())}", flush=True)
xs = torch.stack([c[torch.argsort(c[:, 0])] for c in torch.rand((32, 1024, 3), generator=g)]).cuda()
t2, i2 = run(xs, 20, 2); t1, i1 = run(xs, 20, 1)
print(f"sorted cloud B=32 N=1024 k=20: mfma {t2:7.1f} us   insertion {t1:7.1f} us   equal={bool((i1 == i2).all())}")
