"""conv_f16_kernel at PCN's encoder shapes (B = 64, N = 2048) in its four output modes: fp32 [B,Cout,N], plane image, per-128-point
maxima, image + maxima."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import _fused
from tools.kbench import timeit
g = torch.Generator().manual_seed(0)
B, N = 64, 2048
for Cin, Cout in ((128, 256), (256, 512), (512, 1024)):
    x = torch.randn((B, N, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin), generator=g) / Cin ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    ximg, wimg = _fused.split_rows_f16(x), _fused.split_weights_f16(w)
    gf = 2.0 * B * N * Cin * Cout / 1e6
    for name, fn in (("fp32 out", lambda: _fused.pointwise_conv_f16(ximg, B, N, wimg, Cin, Cout, None, b, relu=True)),
                     ("planes", lambda: _fused.pointwise_conv_f16(ximg, B, N, wimg, Cin, Cout, None, b, relu=True, out_planes=True)),
                     ("pool 128", lambda: _fused.pointwise_conv_f16_pool(ximg, B, N, wimg, Cin, Cout, None, b, relu=True)),
                     ("planes + pool", lambda: _fused.pointwise_conv_f16_pool(ximg, B, N, wimg, Cin, Cout, None, b, relu=True, out_planes=True)),
                     ("group 16", lambda: _fused.pointwise_conv_f16_pool(ximg, B, N, wimg, Cin, Cout, None, b, relu=True, group=16))):
        t = timeit(fn, warm=5, iters=20)
        print(f"{Cin:4d} -> {Cout:4d}  {name:14s} {t:7.1f} us  {gf / t:6.1f} TFLOP/s")
