import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import learning3d_amd.utils as U
g = torch.Generator().manual_seed(0)
for C in (64, 128):
    x = torch.randn((32, C, 1024), generator=g).cuda()
    m = U.knn(x, 20)[:, :, 0].float()
    print(C, "list length mean %.1f  p50 %.0f p99 %.0f max %.0f  (>64: %d)" % (m.mean(), m.median(), m.flatten().kthvalue(int(0.99 * m.numel()))[0], m.max(), int((m > 64).sum())))
