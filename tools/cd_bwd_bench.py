#!/usr/bin/env python3
"""Chamfer backward kernels side by side (raw C-ABI calls): scan of the partner cloud's selections vs the LDS-sorted list."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def main():
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for tag, (B, N, M) in (("c2 B32 1024x1024", (32, 1024, 1024)), ("c4 B8 16384x16384", (8, 16384, 16384)),
                           ("pcn coarse B64 1024x16384", (64, 1024, 16384)), ("B32 2048x2048", (32, 2048, 2048))):
        a, b = torch.rand(B, N, 3, device=dev), torch.rand(B, M, 3, device=dev)
        d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, M, device=dev)
        i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, M, dtype=torch.int32, device=dev)
        check(lib().l3d_chamfer_forward(ptr(a), ptr(b), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd")
        tf = timeit(lambda: check(lib().l3d_chamfer_forward(ptr(a), ptr(b), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd"),
                    warm=2, iters=10)
        g1, g2 = torch.empty_like(a), torch.empty_like(b)
        out = [f"{tag:28s} forward {tf:9.1f} us | backward"]
        for name, v in (("scan", 0), ("sorted", 2)):
            t = timeit(lambda: check(lib().l3d_chamfer_backward_variant(ptr(a), ptr(b), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2),
                                                                        ptr(g1), ptr(g2), v, stream_ptr()), "cd bwd"),
                       warm=2, iters=10)
            out.append(f"{name} {t:9.1f} us")
        print(" ".join(out), flush=True)


if __name__ == "__main__":
    main()
