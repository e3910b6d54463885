"""DCP-v2 forward (BASELINE config 3: B 32 pairs, N 1024, emb 512, 4 heads) -- ms per forward, eager launches, HIP events around 10 forwards.
L3D_TWO_PLANE_IMAGES=0 keeps the pointer network's plane images scaled (rounds 3-5) for an A/B on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import DCP, DGCNN
torch.manual_seed(0)
net = DCP(feature_model=DGCNN(emb_dims=512), cycle=False).cuda().eval()
g = torch.Generator().manual_seed(0)
t = (torch.rand(32, 1024, 3, generator=g) - 0.5).cuda()
s = (t + 0.05).contiguous()
with torch.no_grad():
    for _ in range(3):
        out = net(t, s)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = net(t, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        best = ms if best is None else min(best, ms)
print(f"DCP-v2 forward, two-plane images {os.environ.get('L3D_TWO_PLANE_IMAGES', '1')}: {best:.3f} ms   est_R[0,0] {float(out['est_R'][0, 0, 0]):.6f}")
