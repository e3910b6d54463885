"""Does the Chamfer search (VALU, 71 VGPRs, 36 KB LDS) run beside conv5 (MFMA + DMA, 40 KB of LDS and ~96 registers per SIMD
left) when it sits on a LOWER-priority stream than the DGCNN forward?  hipGraph capture of both variants + eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import DGCNN
from learning3d_amd.losses.chamfer_distance import ChamferDistance, chamfer_loss_local
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).to(dev); a = torch.rand((32, 1024, 3), generator=g).to(dev); b = torch.rand((32, 1024, 3), generator=g).to(dev)
net = DGCNN(emb_dims=1024).to(dev).eval(); cd = ChamferDistance()
hi = torch.cuda.Stream(priority=-1); lo = torch.cuda.Stream(priority=0)
def seq():
    with torch.no_grad():
        f = net(x); d1, d2 = cd(a, b); return f, chamfer_loss_local(d1, d2)
def fork(hi_s, lo_s):
    cur = torch.cuda.current_stream()
    hi_s.wait_stream(cur); lo_s.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(hi_s): f = net(x)
        with torch.cuda.stream(lo_s): d1, d2 = cd(a, b); l = chamfer_loss_local(d1, d2)
    cur.wait_stream(hi_s); cur.wait_stream(lo_s)
    return f, l
def timeit(fn, steps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
def graphed(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): out = fn()
    return gr
print("eager  sequential            %7.1f us" % timeit(seq))
print("eager  fork hi/lo priority   %7.1f us" % timeit(lambda: fork(hi, lo)))
print("eager  fork equal priority   %7.1f us" % timeit(lambda: fork(torch.cuda.Stream(), torch.cuda.Stream())))
for name, fn in (("sequential", seq), ("fork hi/lo priority", lambda: fork(hi, lo)), ("fork equal priority", lambda: fork(torch.cuda.Stream(), torch.cuda.Stream()))):
    try:
        gr = graphed(fn)
        print("graph  %-22s %7.1f us" % (name, timeit(gr.replay)))
    except Exception as e:
        print("graph ", name, "failed:", str(e)[:200])
