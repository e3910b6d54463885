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
    for flag in ((False, False), (True, False), (True, True), (False, False), (True, False), (True, True)):
        F3.FACTOR_FIRST_LAYER, F3.F16_GROUPED_STACK = flag
        t = timeit(lambda: net(pc1, pc2, f1, f2), warm=2, iters=5)
        outs[flag] = net(pc1, pc2, f1, f2)
        print(f"FlowNet3D forward B=32 N=8192, factored first layers {flag[0]!s:5} f16x2 stacks {flag[1]!s:5}: {t:8.1f} us")
    for rows in (16384, 8192):                     # the f16x2 chain only for the largest grouped stacks (fe_layer: 256 x 64 rows per cloud; su3: 1024 x 8)
        F3.FACTOR_FIRST_LAYER, F3.F16_GROUPED_STACK, F3.F16_GROUPED_MIN_ROWS = True, True, rows
        for _ in range(2):
            t = timeit(lambda: net(pc1, pc2, f1, f2), warm=2, iters=5)
            print(f"FlowNet3D forward B=32 N=8192, f16x2 stacks from {rows} rows per cloud: {t:8.1f} us")
        d = (net(pc1, pc2, f1, f2) - outs[(False, False)]).abs().max().item()
        print(f"   max |difference| to the grouped-tensor route: {d:.3e}")
    F3.F16_GROUPED_MIN_ROWS = 0
    for flag in ((True, False), (True, True)):
        d = (outs[flag] - outs[(False, False)]).abs().max().item()
        print(f"max |difference| of {flag} to the grouped-tensor route: {d:.3e} (max |flow| {outs[(False, False)].abs().max().item():.3e})")
F3.FACTOR_FIRST_LAYER, F3.F16_GROUPED_STACK = True, False
