import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd import _lib
if os.environ.get('L3D_LIB'): _lib.LIB_PATH = os.environ['L3D_LIB']
import learning3d_amd.utils as U
from learning3d_amd.models import DGCNN, _fused
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda()
net = DGCNN(emb_dims=64).cuda().eval()
with torch.no_grad():
    idx = U.knn(x.permute(0, 2, 1), 20)
    packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
    a = _fused.edgeconv_forward(x, idx, packed, chained=False)
    for rep in range(1):
        c = _fused.edgeconv_forward(x, idx, packed, chained=True)
        d = (a - c).abs()
        bad = d > 1e-4
        print("rep", rep, "max diff", d.max().item(), "bad frac", bad.float().mean().item())
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): _fused.edgeconv_forward(x, idx, packed, chained=True)
        torch.cuda.synchronize(); print(" chained us:", (time.perf_counter() - t0) / 20 * 1e6)
        if False:
            nz = bad.nonzero()
            print(" first bad", nz[:5].tolist(), "bad per channel-block(64):", [int(bad[..., i*64:(i+1)*64].sum()) for i in range(8)])
            print(" bad points mod 16:", torch.bincount(nz[:, 1] % 16, minlength=16).tolist())
            print(" bad clouds:", torch.bincount(nz[:, 0], minlength=32).tolist())
