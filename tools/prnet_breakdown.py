"""PRNet's dynamic-graph DGCNN (models/prnet.py:62-97) at B=32, N=1024: fused route vs the reference's op
sequence on the same GPU, and the fused route's per-stage times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models.prnet import DGCNN
from learning3d_amd.models import _fused

def timeit(fn, warm=5, iters=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

torch.manual_seed(0)
net = DGCNN(emb_dims=512).cuda().eval()
x = torch.rand(32, 3, 1024, device="cuda")
with torch.no_grad():
    t_f = timeit(lambda: net(x))
net.train()                                  # BatchNorm in train mode -> can_fuse() is False: the reference's op sequence
with torch.no_grad():
    t_r = timeit(lambda: net(x), warm=2, iters=5)
net.eval()
print(f"fused forward        {t_f:8.3f} ms   ({32 / t_f * 1e3:.0f} clouds/s)")
print(f"reference op route   {t_r:8.3f} ms   ({32 / t_r * 1e3:.0f} clouds/s)   speed-up {t_r / t_f:.1f}x")
_fused.TIMER = _fused.StageTimer()
with torch.no_grad():
    for _ in range(10): net(x)
torch.cuda.synchronize()
for k_, v in _fused.TIMER.mean_ms().items():
    n = len(_fused.TIMER.spans[k_]) // 10
    print(f"  {k_:18s} {v * n:8.3f} ms per forward ({n} calls; event-pair timing inflates short kernels by ~15 us each)")
