"""Kernel-level profile of FlowNet3D forward at the c5 per-GPU slice.  Not a product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import FlowNet3D
from torch.profiler import profile, ProfilerActivity
g = torch.Generator().manual_seed(0)
B, N = 32, 8192
pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda(); pc2 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
f1 = torch.rand((B, 3, N), generator=g).cuda(); f2 = torch.rand((B, 3, N), generator=g).cuda()
net = FlowNet3D().cuda().eval()
with torch.no_grad():
    for _ in range(2): net(pc1, pc2, f1, f2)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2): net(pc1, pc2, f1, f2)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=64))
