#!/usr/bin/env python3
"""EdgeConv kernels side by side at the bench shape (B=32, N=1024, k=20): time + error vs the fp32-MFMA kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def main():
    import learning3d_amd.utils as U
    from learning3d_amd.models import DGCNN, _fused
    g = torch.Generator().manual_seed(0)
    B, N, k = 32, 1024, 20
    x = torch.rand((B, N, 3), generator=g).cuda()
    torch.manual_seed(1)
    net = DGCNN(emb_dims=1024).cuda().eval()
    with torch.no_grad():
        idx = U.knn(x.permute(0, 2, 1), k)
        packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        ref = _fused.edgeconv_forward(x, idx, packed, kernel="lds")
        flop = B * N * k * 2 * (6 * 64 + 64 * 64 + 64 * 128 + 128 * 256)
        for kern in sys.argv[1:] or ("f16b", "f16b-planes", "split", "lds"):
            name, planes = (kern[:-7], True) if kern.endswith("-planes") else (kern, False)
            kw = dict(kernel="f16", v2=True) if name == "f16b" else dict(kernel=name)
            if planes:
                kw = dict(planes=True, v2=kw.get("v2", False))
            out = _fused.edgeconv_forward(x, idx, packed, **kw)
            err = float("nan") if planes else (out - ref).abs().max().item()
            for _ in range(3):
                t = timeit(lambda: _fused.edgeconv_forward(x, idx, packed, **kw), warm=20, iters=100)
            print(f"edgeconv {kern:12s} {t:8.1f} us  {flop / t / 1e6:7.1f} TFLOP/s fp32-equiv   max|diff vs chained| {err:.2e}")
        _fused.check_range(x.device, sync=True)


if __name__ == "__main__":
    main()
