#!/usr/bin/env python3
"""Chamfer forward: exact per-pair kernel (variant 2) vs matrix-core ranking + exact refinement (variant 3), us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd._lib import check, lib, ptr, stream_ptr      # noqa: E402
from tools.kbench import timeit                                   # noqa: E402

for B, N in ((32, 1024), (32, 2048), (16, 4096), (8, 16384), (64, 16384)):
    a, b = torch.rand(B, N, 3, device="cuda") - 0.5, torch.rand(B, N, 3, device="cuda") - 0.5
    d1, d2 = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
    i1, i2 = torch.empty(B, N, dtype=torch.int32, device="cuda"), torch.empty(B, N, dtype=torch.int32, device="cuda")
    for v, name in ((2, "packed-exact"), (3, "mfma-ranked")):
        t = timeit(lambda: check(lib().l3d_chamfer_forward_variant(ptr(a), ptr(b), B, N, N, ptr(d1), ptr(d2), ptr(i1), ptr(i2), v, stream_ptr()), "cd"),
                   warm=2, iters=5 if N > 4096 else 30)
        print(f"chamfer_fwd B{B} N{N} {name:13s} {t:10.1f} us  {2.0 * B * N * N / t / 1e3:9.1f} Gpair/s", flush=True)
