#!/usr/bin/env python3
"""pointnet2 ball_query at config 5's shape (32 clouds x 8192 points, 1024 furthest-point-sampled centroids, r 0.5, K 16): the cell-list
kernels against the scanning kernel.  Under rocprofv3 --kernel-trace --stats the two cell-list kernels show separately.  Diagnostic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd.utils import pointnet2_utils as P  # noqa: E402

g = torch.Generator().manual_seed(2000)
xyz = torch.clamp(torch.randn((32, 8192, 3), generator=g), -2, 2).cuda()
fps = P.furthest_point_sample(xyz, 1024)
new = P.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()


def t_us(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for cells in (True, False):
    P.BALL_QUERY_CELLS = cells
    print(f"cells={cells}: {t_us(lambda: P.ball_query(0.5, 16, xyz, new)):.1f} us")
