import sys; sys.path.insert(0, "/root/repo")
import torch, time
from learning3d_amd.models import FlowNet3D, _fused
calls = []
orig_pc, orig_mp = _fused.pointwise_conv, _fused.pointwise_conv_maxpool
def pc(x, w, *a, **k):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    y = orig_pc(x, w, *a, **k); t1.record(); calls.append(("conv", tuple(x.shape), tuple(w.shape), t0, t1)); return y
def mp(x, w, sc, sh, relu, pool, **k):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    y = orig_mp(x, w, sc, sh, relu, pool, **k); t1.record(); calls.append((f"conv+max{pool}", tuple(x.shape), tuple(w.shape), t0, t1)); return y
g = torch.Generator().manual_seed(0); B, N = 32, 8192
pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda(); pc2 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
f1 = torch.rand((B, 3, N), generator=g).cuda(); f2 = torch.rand((B, 3, N), generator=g).cuda()
net = FlowNet3D().cuda().eval()
with torch.no_grad():
    for _ in range(2): net(pc1, pc2, f1, f2)
    _fused.pointwise_conv, _fused.pointwise_conv_maxpool = pc, mp
    import learning3d_amd.models.flownet3d as F3
    net(pc1, pc2, f1, f2); torch.cuda.synchronize()
tot = 0
for name, xs, ws, t0, t1 in calls:
    ms = t0.elapsed_time(t1); tot += ms
    cols = xs[0] * xs[2]
    elig = _fused.split_eligible(ws[1], ws[0], xs[2])
    print(f"{name:12s} x{xs} w{ws} cols={cols:9d} GF={2*ws[0]*ws[1]*cols/1e9:7.1f} {ms*1e3:8.1f} us  bf16x3={elig}")
print("total", tot)
