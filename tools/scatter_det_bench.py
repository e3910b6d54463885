"""Grouping backward at FlowNet3D sa1 size: deterministic (sort + segment sums) vs fp32-atomic scatter."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.utils import pointnet2_utils as P

def timeit(fn, warm=3, iters=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6

B, C, N, S, K = 32, 64, 8192, 1024, 16
xyz = torch.randn(B, N, 3, device="cuda").clamp(-2, 2)
new_xyz = xyz[:, :S].contiguous()
idx = P.ball_query(0.5, K, xyz, new_xyz)
g = torch.randn(B, C, S, K, device="cuda")
from learning3d_amd._lib import lib, check, ptr, stream_ptr
out = torch.empty(B, C, N, device="cuda")
atomic = lambda: check(lib().l3d_group_points_grad(B, C, N, S, K, ptr(g), ptr(idx), ptr(out), stream_ptr()), "a")
det = lambda: P._scatter_add_det(g.view(B, C, S * K), idx, None, N, 1)
print(f"group_points_grad B={B} C={C} N={N} S={S} K={K}: atomic {timeit(atomic):8.1f} us   deterministic {timeit(det):8.1f} us")
