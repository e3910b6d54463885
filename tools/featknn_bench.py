#!/usr/bin/env python3
"""l3d_knn_feature (feature-space kNN, featknn.hip) at B 32 / N 1024 / k 20: us per call incl. the split pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import learning3d_amd.utils as U          # noqa: E402
from tools.kbench import timeit           # noqa: E402

g = torch.Generator().manual_seed(0)
for B, C, N, k in ((32, 64, 1024, 20), (32, 128, 1024, 20), (32, 256, 1024, 20), (32, 64, 2048, 20), (8, 64, 1024, 40)):
    x = torch.randn((B, C, N), generator=g).cuda()
    with torch.no_grad():
        t = timeit(lambda: U.knn(x, k))
    print(f"knn_feature B{B} C{C} N{N} k{k}: {t:8.1f} us   {2.0 * C * B * N * N / t / 1e6:7.1f} TFLOP/s fp32-equiv GEMM", flush=True)
