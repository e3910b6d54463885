import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learning3d_amd.utils as U
from learning3d_amd.models import DGCNN, _fused
from tools.kbench import timeit
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda()
net = DGCNN(emb_dims=1024).cuda().eval()
with torch.no_grad():
    idx = U.knn(x.permute(0, 2, 1), 20)
    packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
    ridx = torch.randint(0, 1024, (32, 1024, 20), generator=g).cuda()
    rpacked = (torch.rand(packed.shape, generator=g) - 0.5).cuda()
    xs = (x - 0.5)
    for name, xx, ii, pp in [("real x, real idx, real w", x, idx, packed), ("real x, rand idx, real w", x, ridx, packed),
                             ("real x, real idx, rand w", x, idx, rpacked), ("centered x, rand idx, rand w", xs, ridx, rpacked),
                             ("real x, real idx, real w", x, idx, packed)]:
        for ch in (True, False):
            t = timeit(lambda: _fused.edgeconv_forward(xx, ii, pp, chained=ch), warm=5, iters=40)
            out = _fused.edgeconv_forward(xx, ii, pp, chained=ch)
            print(f"{name:32s} chained={ch!s:5s} {t:7.1f} us   zeros in pooled: {(out == 0).float().mean().item():.2f}")
