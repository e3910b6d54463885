"""Timeline of featknn_kernel's workgroups (tools/build_variant_lib.py tl featknn.hip -DFK_TIMELINE; L3D_LIB_PATH=tools/bin/libl3d_tl.so):
mean microseconds since the first workgroup's entry at: 0 entry | 1 prologue done | 2 sweep 0's units done | 3 bound known | 4 sweep 1 done | 5 ranked"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd._lib import lib, ptr, stream_ptr, check
B, C, N, k = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1024, 20
x = torch.randn((B, C, N), generator=torch.Generator().manual_seed(0)).cuda()
nb = lib().l3d_knn_feature_workspace_bytes(B, C, N)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
idx = torch.empty((B, N, k), dtype=torch.int64, device="cuda")
for _ in range(5):
    check(lib().l3d_knn_feature(ptr(x), B, C, N, k, ptr(ws), ptr(idx), stream_ptr()), "knn")
torch.cuda.synchronize()
Cp = (C + 63) // 64 * 64
off = B * Cp * N * 4 + B * N * 8
marks = ws[off:off + B * (N // 128) * 64].view(torch.int64).view(-1, 8).cpu().double()
t0 = marks[:, 0].min()
us = (marks - t0) / 100.0
names = ["entry", "prologue done", "sweep 0 units done", "bound known", "sweep 1 done", "ranked"]
for i, n in enumerate(names):
    print(f"  {n:20s} mean {us[:, i].mean():7.2f} us   min {us[:, i].min():7.2f}   max {us[:, i].max():7.2f}")
