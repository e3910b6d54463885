"""Micro-timing of the transformer's building blocks at c3 shapes.  Not a product path."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from learning3d_amd.utils.transformer import LayerNorm
def timeit(fn, warm=3, iters=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, H, N, D = 32, 4, 1024, 128
q = torch.randn(B, H, D, N, device="cuda"); k = torch.randn(B, H, D, N, device="cuda"); v = torch.randn(B, H, D, N, device="cuda")
x = torch.randn(B, N, 512, device="cuda")
ln = LayerNorm(512).cuda()
with torch.no_grad():
    print("QK^T (q^T k)        %8.1f us" % timeit(lambda: torch.matmul(q.transpose(-2, -1), k)))
    s = torch.matmul(q.transpose(-2, -1), k)
    print("scale               %8.1f us" % timeit(lambda: s / math.sqrt(D)))
    print("softmax             %8.1f us" % timeit(lambda: F.softmax(s, dim=-1)))
    p = F.softmax(s, dim=-1)
    print("PV (v p^T)          %8.1f us" % timeit(lambda: torch.matmul(v, p.transpose(-2, -1))))
    print("LayerNorm (ref ops) %8.1f us" % timeit(lambda: ln(x)))
    print("residual add (strided view) %8.1f us" % timeit(lambda: x + v.view(B, 512, N).transpose(1, 2)))
    qq = q.transpose(-2, -1).contiguous()
    print("sdpa fused          %8.1f us" % timeit(lambda: F.scaled_dot_product_attention(qq, k.transpose(-2, -1).contiguous(), v.transpose(-2, -1).contiguous())))
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    qc, kc, vc = [z.reshape(B, H * D, N).contiguous() for z in (q, k, v)]
    out = torch.empty_like(qc)
    def flash():
        check(lib().l3d_attention_forward_strided(ptr(qc), ptr(kc), ptr(vc), B, H, D, N, N, H * D * N, H * D * N, H * D * N, 1 / math.sqrt(D), ptr(out), stream_ptr()), "att")
    print("flash attention (l3d)  %8.1f us" % timeit(flash))
    from learning3d_amd.models import _fused
    w = torch.randn(512, 512, device="cuda"); bias = torch.randn(512, device="cuda")
    ws = _fused.split_rows(w)
    print("linear 512->512 (conv_split, cl in) %8.1f us" % timeit(lambda: _fused.pointwise_conv(x, w, None, bias, channel_last=True, w_split=ws)))
    xcf = x.transpose(1, 2).contiguous()
    print("linear 512->512 (conv_split, cf in) %8.1f us" % timeit(lambda: _fused.pointwise_conv(xcf, w, None, bias, channel_last=False, w_split=ws)))
    print("linear 512->512 (torch)             %8.1f us" % timeit(lambda: F.linear(x, w, bias)))
