"""Times l3d_knn_variant's kernels (1 = lane per query, knn.hip; 2 = wave per query, knn_select.hip; 3 = four slots, knn_small.hip)
on a few shapes.  `python tools/knn_select_bench.py small`: only the k <= 4 shapes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from learning3d_amd._lib import check, lib, ptr, stream_ptr


def timeit(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    small = [(32, 8192, 1024, 3, "normal"), (32, 1024, 1024, 3, "normal"), (32, 1024, 8192, 4, "normal"), (32, 256, 64, 3, "normal"),
             (32, 1024, 256, 3, "normal"), (64, 8192, 1024, 3, "normal"), (32, 2048, 2048, 1, "normal")]
    for (B, n, m, k, kind) in small if "small" in sys.argv else [(32, 1024, 8192, 64, "normal"), (32, 1024, 8192, 64, "sorted"), (32, 1024, 8192, 64, "grid"),
                               (32, 256, 256, 64, "normal"), (32, 1024, 8192, 128, "normal"), (32, 1024, 8192, 200, "normal"),
                               (32, 1024, 2048, 64, "normal"), (32, 1024, 8192, 40, "normal"), (32, 1024, 8192, 32, "normal"),
                               (32, 1024, 8192, 16, "normal"), (32, 1024, 8192, 8, "normal"), (32, 1024, 1024, 32, "normal"),
                               (32, 1024, 1024, 16, "normal"), (32, 256, 64, 8, "normal"), (32, 8192, 1024, 3, "normal"), (32, 1024, 256, 8, "normal"),
                               (32, 1024, 8192, 4, "normal"), (32, 2048, 2048, 16, "normal"), (32, 1024, 512, 16, "normal"),
                               (32, 1024, 1024, 8, "normal"), (32, 1024, 1024, 3, "normal"), (4, 1024, 8192, 16, "normal")]:
        c = torch.clamp(torch.randn((B, m, 3), generator=g), -2, 2)
        if kind == "sorted":
            c = torch.stack([x[torch.argsort(x[:, 0])] for x in c])
        if kind == "grid":
            c = torch.round(c * 2) / 2
        c = c.cuda()
        q = (c[:, torch.randperm(m, generator=g)[:n]].contiguous() if n <= m
             else torch.clamp(torch.randn((B, n, 3), generator=g), -2, 2).cuda())
        d = torch.empty((B, n, k), device="cuda")
        i = torch.empty((B, n, k), dtype=torch.int32, device="cuda")
        ts = []
        for v in (1, 2) + ((3,) if k <= 4 else ()):
            ts.append(timeit(lambda: check(lib().l3d_knn_variant(B, n, m, k, ptr(q), ptr(c), ptr(d), ptr(i), v, stream_ptr()), "knn")))
        print(f"B {B} n {n} m {m} k {k} {kind:7s}: lane kernel {ts[0]:9.1f} us   select kernel {ts[1]:9.1f} us"
              + (f"   four-slot kernel {ts[2]:9.1f} us" if k <= 4 else ""), flush=True)


if __name__ == "__main__":
    main()
