"""Why is the EdgeConv kernel 5-12 % slower inside the step than back to back?  Its duration (events around the launch)
after different predecessors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learning3d_amd.utils as U
from learning3d_amd.models import DGCNN, _fused
from learning3d_amd.losses.chamfer_distance import ChamferDistance
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda(); a = torch.rand((32, 1024, 3), generator=g).cuda(); b = torch.rand((32, 1024, 3), generator=g).cuda()
net = DGCNN(emb_dims=1024).cuda().eval(); cd = ChamferDistance()
with torch.no_grad():
    xt = x.permute(0, 2, 1)
    idx = U.knn(xt, 20)
    packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
    w5, s5, b5, w5s, w5f = net._conv5_folded()
    img = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True)
    big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
    pre = {"nothing": lambda: None, "knn": lambda: U.knn(xt, 20), "conv5": lambda: _fused.pointwise_conv_f16(img, 32, 1024, w5f, 512, 1024, s5, b5, relu=True),
           "chamfer": lambda: cd(a, b), "1 GB memset": lambda: big.zero_(),
           "whole step order (chamfer, knn)": lambda: (cd(a, b), U.knn(xt, 20))}
    for name, fn in pre.items():
        ts = []
        for it in range(30):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            timer = _fused.StageTimer(only=("edgeconv_kernel",)); _fused.TIMER = timer
            _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True)
            _fused.TIMER = None
            torch.cuda.synchronize()
            ts.append(list(timer.mean_ms().values())[0] * 1e3)
        ts = sorted(ts[5:])
        print(f"after {name:34s}: median {ts[len(ts)//2]:7.1f} us   min {ts[0]:7.1f}")
