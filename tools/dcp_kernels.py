"""Kernel-level profile of DCP forward at BASELINE config 3 (B=32, N=1024, emb 512).  Not a product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import DCP, DGCNN
from torch.profiler import profile, ProfilerActivity
g = torch.Generator().manual_seed(0)
B, N = 32, 1024
t = (torch.rand((B, N, 3), generator=g) - 0.5).cuda(); s = (torch.rand((B, N, 3), generator=g) - 0.5).cuda()
net = DCP(feature_model=DGCNN(emb_dims=512), cycle=False).cuda().eval()
with torch.no_grad():
    for _ in range(3): net(t, s)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2): net(t, s)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=24, max_name_column_width=64))
