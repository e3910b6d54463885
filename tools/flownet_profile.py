"""Kernel table of one FlowNet3D forward at config 5's per-GPU shape (B 32, N 8192), torch profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import learning3d_amd.models.flownet3d as F3
torch.manual_seed(0)
net = F3.FlowNet3D().cuda().eval()
g = torch.Generator().manual_seed(3)
B, N = 32, 8192
pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
pc2 = (pc1 + 0.05 * torch.randn((B, 3, N), generator=g).cuda()).contiguous()
f1 = torch.rand((B, 3, N), generator=g).cuda(); f2 = torch.rand((B, 3, N), generator=g).cuda()
iters = 3
with torch.no_grad():
    for _ in range(2):
        net(pc1, pc2, f1, f2)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            net(pc1, pc2, f1, f2)
        torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(k.self_device_time_total for k in ka) / iters / 1e3
print(f"FlowNet3D forward: {tot:.2f} ms of kernels in {sum(k.count for k in ka) / iters:.0f} launches")
for k in sorted(ka, key=lambda k: -k.self_device_time_total)[:24]:
    print(f"   {k.self_device_time_total / iters / 1e3:8.3f} ms  x{k.count / iters:5.1f}  {k.key[:110]}")
