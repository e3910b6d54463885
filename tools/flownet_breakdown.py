"""FlowNet3D forward at the c5 per-GPU slice (B=32, N=8192): where the time goes.  Not a product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import FlowNet3D
def timeit(fn, warm=2, iters=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator().manual_seed(0)
B, N = 32, 8192
pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
pc2 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
f1 = torch.rand((B, 3, N), generator=g).cuda(); f2 = torch.rand((B, 3, N), generator=g).cuda()
net = FlowNet3D().cuda().eval()
with torch.no_grad():
    print("FlowNet3D forward  %9.1f us" % timeit(lambda: net(pc1, pc2, f1, f2)))
    print("sa1                %9.1f us" % timeit(lambda: net.sa1(pc1, f1)))
    l1p, l1f = net.sa1(pc1, f1)
    print("sa2                %9.1f us" % timeit(lambda: net.sa2(l1p, l1f)))
    l2p, l2f = net.sa2(l1p, l1f)
    l1p2, l1f2 = net.sa1(pc2, f2); l2p2, l2f2 = net.sa2(l1p2, l1f2)
    print("flow embedding     %9.1f us" % timeit(lambda: net.fe_layer(l2p, l2p2, l2f, l2f2)))
    _, l2n = net.fe_layer(l2p, l2p2, l2f, l2f2)
    print("sa3                %9.1f us" % timeit(lambda: net.sa3(l2p, l2n)))
    l3p, l3f = net.sa3(l2p, l2n)
    print("sa4                %9.1f us" % timeit(lambda: net.sa4(l3p, l3f)))
    l4p, l4f = net.sa4(l3p, l3f)
    print("su1                %9.1f us" % timeit(lambda: net.su1(l3p, l4p, l3f, l4f)))
    l3n = net.su1(l3p, l4p, l3f, l4f)
    print("su2                %9.1f us" % timeit(lambda: net.su2(l2p, l3p, torch.cat([l2f, l2n], dim=1), l3n)))
    l2nn = net.su2(l2p, l3p, torch.cat([l2f, l2n], dim=1), l3n)
    print("su3                %9.1f us" % timeit(lambda: net.su3(l1p, l2p, l1f, l2nn)))
    l1n = net.su3(l1p, l2p, l1f, l2nn)
    print("fp                 %9.1f us" % timeit(lambda: net.fp(pc1, l1p, f1, l1n)))
