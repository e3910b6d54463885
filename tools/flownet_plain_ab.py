"""FlowNet3D forward at config 5's per-GPU shape (B 32, N 8192): the per-point stacks (feature propagation, head conv1) as an f16x2 chain
(F16_PLAIN_STACK) against the bf16x3 / fp32-MFMA layers, interleaved on one box, and the difference between the two outputs."""
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
outs = {}
with torch.no_grad():
    for rep in range(3):
        for flag in (False, True):
            F3.F16_PLAIN_STACK = flag
            t = timeit(lambda: net(pc1, pc2, f1, f2), warm=2, iters=5)
            outs[flag] = net(pc1, pc2, f1, f2)
            print(f"FlowNet3D forward B=32 N=8192, f16x2 per-point stacks {flag!s:5}: {t:8.1f} us")
d = (outs[True] - outs[False]).abs().max().item()
print(f"max |difference| between the two: {d:.3e} (max |flow| {outs[False].abs().max().item():.3e})")
