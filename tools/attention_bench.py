"""l3d_attention_forward at DCP's shape (B=32, H=4, D=128, N=M=1024): time per call, fp32-equivalent TFLOP/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning3d_amd._lib import lib, check, ptr, stream_ptr
B, H, D, N = 32, 4, 128, 1024
q, k, v = (torch.randn(B, H * D, N, device="cuda") for _ in range(3))
ctx = torch.empty_like(q)
fn = lambda: check(lib().l3d_attention_forward_strided(ptr(q), ptr(k), ptr(v), B, H, D, N, N, H * D * N, H * D * N, H * D * N, 1.0 / D ** 0.5, ptr(ctx), stream_ptr()), "att")
for _ in range(10): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): fn()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"attention B={B} H={H} D={D} N={N}: {dt*1e6:.1f} us  {4.0*B*H*N*N*D/dt/1e12:.1f} TFLOP/s fp32-equiv")
ws = torch.zeros(4, dtype=torch.int32, device="cuda")
img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, H * D), dtype=torch.uint8, device="cuda")
for name, kw in (("f16x2 restructured (attention_f16b), fp32 ctx, incl. absmax", dict(mr=0, c=ctx, im=None)),
                 ("f16x2 restructured, maxima ready, plane image out (DCP's call)", dict(mr=1, c=None, im=img))):
    fnb = lambda: check(lib().l3d_attention_forward_f16b(ptr(q), ptr(k), ptr(v), B, H, D, N, N, H * D * N, H * D * N, H * D * N, 1.0 / D ** 0.5,
                                                         ptr(ws), kw["mr"], ptr(kw["c"]), ptr(kw["im"]), stream_ptr()), "att16b")
    for _ in range(10): fnb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fnb()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"attention {name}: {dt*1e6:.1f} us  {4.0*B*H*N*N*D/dt/1e12:.1f} TFLOP/s fp32-equiv")
