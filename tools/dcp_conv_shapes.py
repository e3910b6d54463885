"""conv_f16_kernel at the shapes and output modes of DCP-v2's pointer network (B 32 pairs, N 1024, emb 512, ff 1024): q|k|v, q, k|v
projections with the operand maxima, output projection and FF w_2 with the residual epilogue, FF w_1 with ReLU into a plane image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd.models import _fused
from tools.kbench import timeit
g = torch.Generator().manual_seed(0)
B, N = 32, 1024
ws = torch.zeros(4, dtype=torch.int32, device="cuda")
for name, Cin, Cout, mode in (("q|k|v  absmax", 512, 1536, "amax"), ("k|v    absmax", 512, 1024, "amax"), ("q      absmax", 512, 512, "amax"),
                              ("out    residual", 512, 512, "res"), ("ff w_2 residual", 1024, 512, "res"), ("ff w_1 planes", 512, 1024, "planes"),
                              ("plain fp32 512->512", 512, 512, "plain"), ("plain fp32 512->1024", 512, 1024, "plain")):
    x = torch.randn((B, N, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin), generator=g) / Cin ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    res = torch.randn((B, Cout, N), generator=g).cuda()
    ximg, wimg = _fused.split_rows_f16(x), _fused.split_weights_f16(w)
    gf = 2.0 * B * N * Cin * Cout / 1e6
    ts = []
    for two in (False, True):       # the image's residual plane scaled by 2^12 (three weight planes) / unscaled (two)
        fn = {"amax": lambda: _fused.pointwise_conv_f16(ximg, B, N, wimg, Cin, Cout, None, b, amax=(ws, 512), unscaled=two),
              "res": lambda: _fused.pointwise_conv_f16(ximg, B, N, wimg, Cin, Cout, None, b, residual=res, unscaled=two),
              "planes": lambda: _fused.pointwise_conv_f16(ximg, B, N, wimg, Cin, Cout, None, b, relu=True, out_planes=True, unscaled=two),
              "plain": lambda: _fused.pointwise_conv_f16(ximg, B, N, wimg, Cin, Cout, None, b, unscaled=two)}[mode]
        ts.append(timeit(fn, warm=5, iters=20))
    print(f"{name:22s} {Cin:4d} -> {Cout:4d}  three planes {ts[0]:7.1f} us {gf / ts[0]:6.1f} TF   two planes {ts[1]:7.1f} us {gf / ts[1]:6.1f} TF (fp32-equivalent)")
