"""FlowNet3D forward at BASELINE config 5's per-GPU shape (B=32, N=8192), with and without the factored first layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learning3d_amd.models.flownet3d as F3
from tools.kbench import timeit
torch.manual_seed(0)
net = F3.FlowNet3D().cuda().eval()
g = torch.Generator().manual_seed(3)
B, N = 32, 8192
pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
pc2 = (pc1 + 0.05 * torch.randn((B, 3, N), generator=g).cuda()).contiguous()
f1 = torch.rand((B, 3, N), generator=g).cuda(); f2 = torch.rand((B, 3, N), generator=g).cuda()
with torch.no_grad():
    outs = {}
    for flag in (False, True, False, True):
        F3.FACTOR_FIRST_LAYER = flag
        t = timeit(lambda: net(pc1, pc2, f1, f2), warm=2, iters=5)
        outs[flag] = net(pc1, pc2, f1, f2)
        print(f"FlowNet3D forward B=32 N=8192, factored first layers {flag!s:5}: {t:8.1f} us")
    d = (outs[True] - outs[False]).abs().max().item()
    print(f"max |difference| between the routes: {d:.3e} (max |flow| {outs[False].abs().max().item():.3e})")
