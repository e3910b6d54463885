"""PCN forward at BASELINE config 4's shape (B=64, 2048 -> 16384 points) in the two matrix-core arithmetics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import PCN, _fused
from tools.kbench import timeit
g = torch.Generator().manual_seed(0)
pcn = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).cuda().eval()
part = (torch.rand((64, 2048, 3), generator=g) - 0.5).cuda()
with torch.no_grad():
    for arith in ("f16x2", "bf16x3"):
        _fused.GEMM_ARITH = arith
        t = timeit(lambda: pcn(part), warm=5, iters=10)
        print(f"PCN forward B=64, {arith}: {t:8.1f} us   {64 / t * 1e6:8.0f} clouds/s")
