"""feature-space kNN a few times (for rocprofv3 --kernel-trace --stats): B 32, N 1024, k 20; C from argv (default 64)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import learning3d_amd.utils as U
g = torch.Generator().manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn((32, C, 1024), generator=g).cuda()
for _ in range(5):
    U.knn(x, 20)
torch.cuda.synchronize()
