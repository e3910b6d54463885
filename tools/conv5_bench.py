#!/usr/bin/env python3
"""conv5 (512 -> 1024 over 32 x 1024 points) in the three arithmetics: time and TFLOP/s (fp32-equivalent)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def main():
    from learning3d_amd.models import _fused
    g = torch.Generator().manual_seed(0)
    B, N, Cin, Cout = 32, 1024, 512, 1024
    x = torch.relu(torch.randn((B, N, Cin), generator=g)).cuda()
    w = (torch.randn((Cout, Cin), generator=g) / Cin ** 0.5).cuda()
    sc, sh = torch.rand(Cout, generator=g).cuda() + 0.5, torch.rand(Cout, generator=g).cuda()
    flop = 2.0 * B * N * Cin * Cout
    with torch.no_grad():
        ref = _fused.pointwise_conv(x, w, sc, sh, relu=True, channel_last=True, split=False)
        ws = _fused.split_rows(w)
        wp = _fused.split_weights_f16(w)
        xp = _fused.split_rows_f16(x)
        runs = {
            "fp32-mfma": lambda: _fused.pointwise_conv(x, w, sc, sh, relu=True, channel_last=True, split=False),
            "bf16x3": lambda: _fused.pointwise_conv(x, w, sc, sh, relu=True, channel_last=True, w_split=ws, split=True),
            "f16x2 (pre-split x)": lambda: _fused.pointwise_conv_f16(xp, B, N, wp, Cin, Cout, sc, sh, relu=True),
            "f16x2 two-plane (time only)": lambda: _fused.pointwise_conv_f16(xp, B, N, wp, Cin, Cout, sc, sh, relu=True, unscaled=True),   # xp's residual is the scaled one: values are off, the kernel's time is not
            "f16x2 + x split pass": lambda: _fused.pointwise_conv_f16(_fused.split_rows_f16(x), B, N, wp, Cin, Cout, sc, sh, relu=True),
        }
        for name, fn in runs.items():
            err = (fn() - ref).abs().max().item()
            for _ in range(2):
                t = timeit(fn, warm=20, iters=100)
            print(f"conv5 {name:22s} {t:8.1f} us  {flop / t / 1e6:7.1f} TFLOP/s fp32-equiv   max|diff vs fp32-mfma| {err:.2e}")
        _fused.check_range(x.device, sync=True)


if __name__ == "__main__":
    main()
