import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from learning3d_amd._lib import lib, check, ptr, stream_ptr
from learning3d_amd.models import _fused
B, Cin, Cout, N = 32, 512, 1024, 1024
x = torch.randn(B, Cin, N, device="cuda"); w = torch.randn(Cout, Cin, device="cuda") / 22
ws = _fused.split_rows(w)
xt = x.transpose(1, 2).contiguous().view(B * N, Cin)
xs = _fused.split_rows(xt)
y = torch.empty(B, Cout, N, device="cuda")
def run(mode, xin):
    return lambda: check(lib().l3d_pointwise_conv_split(ptr(xin), mode, ptr(ws), None, None, 0, B, Cin, Cout, N, 1, 0, ptr(y), stream_ptr()), "c")
for name, fn in (("x_mode 0 (fp32 channel-first)", run(0, x)), ("x_mode 2 (pre-split)", run(2, xs))):
    for _ in range(600): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize(); print(name, f"{(time.perf_counter()-t0)/200*1e6:.1f} us")
