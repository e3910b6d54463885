#!/usr/bin/env python3
"""Where do two EdgeConv variants' plane images differ?  usage: python tools/lab_diff.py <good variant> <other variant>"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import learning3d_amd.utils as U
from learning3d_amd._lib import ptr, stream_ptr
from learning3d_amd.models import DGCNN, _fused
BIN = os.path.join(ROOT, "tools", "bin")
g = torch.Generator().manual_seed(1000)
B, N, k = 32, 1024, 20
x = torch.rand((B, N, 3), generator=g).cuda()
torch.manual_seed(1)
net = DGCNN(emb_dims=1024).cuda().eval()
with torch.no_grad():
    idx = U.knn(x.permute(0, 2, 1), k)
    packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
flag = _fused.range_flag(x.device)
nbytes = 2 * 512 * B * N * 2 + 16
outs = []
for n in sys.argv[1:3]:
    L = ctypes.CDLL(os.path.join(BIN, f"libef_{n}.so"))
    fn = L.l3d_edgeconv_forward_f16b
    fn.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 2
    out = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    assert fn(ptr(x), ptr(idx), B, N, k, ptr(packed), ptr(out), 2, ptr(flag), stream_ptr()) == 0
    torch.cuda.synchronize()
    outs.append(out[:nbytes - 16].view(torch.int16).cpu().numpy().reshape(2, 64, B * N, 8))      # [plane][cell row][point][ch & 7]
a, b = outs
d = a != b
print("differing halves:", int(d.sum()), "of", d.size)
for pl in range(2):
    per_ch = d[pl].transpose(0, 2, 1).reshape(512, B * N).sum(1)             # [cell row][ch&7][point] -> channel = 8 row + c
    nz = np.nonzero(per_ch)[0]
    print(f"plane {pl}: channels with differences: {len(nz)}; first {nz[:12].tolist()} last {nz[-12:].tolist()}")
    for lo, hi in ((0, 32), (32, 64), (64, 128), (128, 256), (256, 512)):
        print(f"   channels {lo:3d}..{hi - 1:3d}: {int(per_ch[lo:hi].sum()):9d} differing of {(hi - lo) * B * N}")
per_pt = d.reshape(2 * 64, B * N, 8).sum((0, 2)).reshape(-1, 16).sum(0)
print("by point position inside a 16-point tile:", per_pt.tolist())
fa = a.view(np.float16).astype(np.float32); fb = b.view(np.float16).astype(np.float32)
sel = d[0]
print("h plane: sample good/other values where they differ:", fa[0][sel][:8].tolist(), fb[0][sel][:8].tolist())
