"""PCN forward at c4 (B=64, 2048 -> 16384 points): stage timing via _fused.StageTimer spans if present, else whole."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import PCN, _fused
g = torch.Generator().manual_seed(0)
x = (torch.rand((64, 2048, 3), generator=g) - 0.5).cuda()
net = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).cuda().eval()
def timeit(fn, warm=2, iters=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
with torch.no_grad():
    print("PCN forward %.1f us" % timeit(lambda: net(x)))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3): net(x)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
