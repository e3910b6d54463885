#!/usr/bin/env python3
"""l3d_bmm_f32 against torch.matmul (rocBLAS) on the shapes of a DCP-v2 training step (B 8, N 1024, emb 512, 4 heads).  Diagnostic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd.models import _rows  # noqa: E402


def t_us(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


g = torch.Generator().manual_seed(0)
NB = int(os.environ.get("DCP_B", "8"))
R = NB * 1024
x, w, gy = torch.randn(R, 512, generator=g).cuda(), torch.randn(512, 512, generator=g).cuda(), torch.randn(R, 512, generator=g).cuda()
w2 = torch.randn(1024, 512, generator=g).cuda()
q = torch.randn(NB, 1024, 4, 128, generator=g).cuda().transpose(1, 2)
k = torch.randn(NB, 1024, 4, 128, generator=g).cuda().transpose(1, 2)
p = torch.randn(NB, 4, 1024, 1024, generator=g).cuda()
cases = [
    ("linear fwd  x W^T (w.t() view)      Rx512x512", x, w.t(), 1),
    ("linear fwd  x W^T (own tensor: the route)  Rx512x512", x, w.t().contiguous(), 1),
    ("linear fwd  x W2^T (own tensor)  Rx1024x512", x, w2.t().contiguous(), 1),
    ("dgrad       g W          Rx512x512", gy, w, 1),
    ("wgrad       g^T x        512x512xR split", gy.t(), x, _rows._split_parts(512, 512, R)),
    ("q k^T       32 x 1024x1024x128", q, k.transpose(-1, -2), 1),
    ("p v         32 x 1024x128x1024", p, k, 1),
    ("p^T dO      32 x 1024x128x1024", p.transpose(-1, -2), q, 1),
    ("dS^T q", p.transpose(-1, -2), q, 1),
]
tc = t_us(lambda: _rows.colsum(gy))
tt = t_us(lambda: gy.sum(0))
print(f"{'bias grad  colsum(g) (l3d_colsum_rows)  Rx512':48s} hip {tc:8.1f} us   torch sum(0) {tt:8.1f} us", flush=True)
for name, a, b, parts in cases:
    flop = 2.0 * a.shape[-2] * a.shape[-1] * b.shape[-1] * (a.numel() // (a.shape[-2] * a.shape[-1]))
    th = t_us(lambda: _rows.bmm(a, b, parts=parts))
    tt = t_us(lambda: torch.matmul(a, b))
    print(f"{name:48s} bmm {th:8.1f} us {flop / th / 1e6:6.1f} TF   torch {tt:8.1f} us {flop / tt / 1e6:6.1f} TF   ratio {th / tt:4.2f}", flush=True)
