import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learning3d_amd.models import DGCNN, _fused
import learning3d_amd.utils as U
torch.manual_seed(4)
net = DGCNN(emb_dims=64).cuda().eval()
for (B, N, k) in [(1, 50, 7), (2, 77, 16), (2, 128, 20), (1, 64, 8), (1, 64, 4)]:
    x = torch.rand(B, N, 3, device="cuda")
    with torch.no_grad():
        idx = U.knn(x.permute(0, 2, 1), k)
        packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        c = _fused.edgeconv_forward(x, idx, packed, kernel="chained").cpu().numpy()
        s = _fused.edgeconv_forward(x, idx, packed, kernel="split").cpu().numpy()
    d = np.abs(c - s)
    print((B, N, k), "max diff per layer block:", [float(d[..., a:b].max()) for a, b in ((0, 64), (64, 128), (128, 256), (256, 512))],
          "worst point idx", np.unravel_index(d.argmax(), d.shape))
