import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import learning3d_amd.utils as U
g = torch.Generator().manual_seed(0)
x = torch.randn((32, 64, 1024), generator=g).cuda()
for _ in range(5):
    U.knn(x, 20)
torch.cuda.synchronize()
