"""Driver for a rocprofv3 --pmc pass: the two round-3 kNN kernels at config 5's shapes, five launches each
(knn_select_kernel<128>: k = 64, 1024 queries x 8192 candidates x 32 clouds; knn_small_kernel: three_nn, 8192 queries x 1024 candidates)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.utils import pointnet2_utils as P
g = torch.Generator().manual_seed(0)
xyz = torch.clamp(torch.randn((32, 8192, 3), generator=g), -2, 2).cuda()
new_xyz = xyz[:, :1024].contiguous()
for _ in range(5):
    P.knn(64, new_xyz, xyz)
    P.three_nn(xyz, new_xyz)
torch.cuda.synchronize()
