"""How many 32-query blocks of the matrix-core kNN overflow their candidate lists (and go to the fix-up kernel)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd._lib import check, lib, ptr, stream_ptr
g = torch.Generator().manual_seed(0)
B, N, k = 32, 1024, 20
sets = {"U(0,1)": torch.rand((B, N, 3), generator=g), "U(-.5,.5)": torch.rand((B, N, 3), generator=g) - 0.5,
        "U(-.5,.5) b": torch.rand((B, N, 3), generator=g) - 0.5, "N(0,1)": torch.randn((B, N, 3), generator=g),
        "U(-1,1)": torch.rand((B, N, 3), generator=g) * 2 - 1, "U(0,1)+10": torch.rand((B, N, 3), generator=g) + 10}
for name, x in sets.items():
    x = x.cuda().contiguous()
    idx = torch.empty((B, N, k), dtype=torch.int64, device="cuda")
    check(lib().l3d_knn_graph_variant(ptr(x), B, N, k, ptr(idx), 3, stream_ptr()), "knn")
    torch.cuda.synchronize()
    marked = (idx[:, ::32, 0] == -1)
    print(f"{name:14s} overflowed blocks: {int(marked.sum())} of {marked.numel()}   per cloud: {marked.sum(1).tolist()[:8]}")
