#!/usr/bin/env python3
"""EMD forward / backward: the round-5 sweep kernels (emd.hip) against the retired one-workgroup-per-cloud kernel
(tools/experiments/emd_v1.hip -> tools/bin/libemd_v1.so, built with the product flags) and the reference's own kernels
(oracle/_ref/libref_emd_nofma.so): bit comparison of match / cost, then HIP-event timing at the BASELINE shapes.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared tools/experiments/emd_v1.hip -o tools/bin/libemd_v1.so
Not a product path."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning3d_amd._lib import check, lib, stream_ptr            # noqa: E402
from tools.kbench import timeit                                    # noqa: E402

p = lambda t: C.c_void_p(t.data_ptr())


def load(path):
    return C.CDLL(os.path.join(ROOT, path)) if os.path.exists(os.path.join(ROOT, path)) else None


def main():
    V1, REF = load("tools/bin/libemd_v1.so"), load("oracle/_ref/libref_emd_nofma.so")
    L = lib()
    rng = np.random.default_rng(11)
    print("== bits: new (split 1/2/4) vs v1 vs the reference kernel (-ffp-contract=off)")
    for B, n, m in ((2, 256, 256), (4, 1024, 1024), (2, 300, 900), (2, 512, 256), (3, 1000, 1000), (2, 2048, 2048)):
        a = torch.from_numpy(rng.uniform(0, 1, (B, n, 3)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.uniform(0, 1, (B, m, 3)).astype(np.float32)).cuda()
        ws = torch.empty(L.l3d_emd_workspace_bytes(B, n, m), dtype=torch.uint8, device="cuda")
        outs = {}
        for split in (1, 2, 4):
            match = torch.full((B, m, n), float("nan"), device="cuda"); cost = torch.empty(B, device="cuda")
            check(L.l3d_emd_forward(p(a), p(b), B, n, m, p(match), p(cost), p(ws), split, stream_ptr()), "emd")
            outs[f"s{split}"] = (match, cost)
        if V1 is not None:
            match = torch.empty((B, m, n), device="cuda"); cost = torch.empty(B, device="cuda")
            temp = torch.empty((B, 2 * (n + m)), device="cuda")
            assert V1.l3d_emd_forward_v1(p(a), p(b), B, n, m, p(match), p(cost), p(temp), stream_ptr()) == 0
            outs["v1"] = (match, cost)
        if REF is not None:
            match = torch.zeros((B, m, n), device="cuda"); cost = torch.zeros(B, device="cuda")
            temp = torch.zeros((B, 2 * (n + m)), device="cuda")
            torch.cuda.synchronize()
            REF.ref_emd_forward(B, n, m, p(a), p(b), p(match), p(temp), p(cost))
            outs["ref"] = (match, cost)
        torch.cuda.synchronize()
        base = outs["s2"]
        line = [f"B{B} n{n} m{m}:"]
        for k, (mt, c) in outs.items():
            if k == "s2":
                continue
            dm = float((mt - base[0]).abs().max()); eq = bool(torch.equal(mt, base[0]))
            dc = float(((c - base[1]).abs() / base[1].abs()).max())
            line.append(f"{k}: match {'BIT-EQUAL' if eq else f'maxdiff {dm:.3e}'} cost rel {dc:.2e};")
        if REF is not None:                                 # backward on the reference's match: gradients bit for bit?
            wm = outs["ref"][0]
            g1, g2 = torch.empty_like(a), torch.empty_like(b)
            check(L.l3d_emd_backward(p(a), p(b), p(wm), B, n, m, p(g1), p(g2), stream_ptr()), "emd bwd")
            w1, w2 = torch.zeros_like(a), torch.zeros_like(b)
            torch.cuda.synchronize()
            REF.ref_emd_backward(B, n, m, p(a), p(b), p(wm), p(w1), p(w2))
            torch.cuda.synchronize()
            for nm, got, want in (("grad1", g1, w1), ("grad2", g2, w2)):
                line.append(f"{nm} {'BIT-EQUAL' if torch.equal(got, want) else f'maxdiff {float((got - want).abs().max()):.3e} of {float(want.abs().max()):.2e}'};")
        print(" ".join(line), flush=True)

    print("== time (us per call, best of 3 batches)")
    for B, n in ((32, 1024), (64, 1024), (8, 2048), (256, 1024)):
        m = n
        a = torch.rand(B, n, 3, device="cuda"); b = torch.rand(B, m, 3, device="cuda")
        ws = torch.empty(L.l3d_emd_workspace_bytes(B, n, m), dtype=torch.uint8, device="cuda")
        match = torch.empty((B, m, n), device="cuda"); cost = torch.empty(B, device="cuda")
        pairs = B * n * m
        for split in (0, 1, 2, 4):
            t = timeit(lambda: check(L.l3d_emd_forward(p(a), p(b), B, n, m, p(match), p(cost), p(ws), split, stream_ptr()), "emd"),
                       warm=2, iters=10)
            print(f"emd_fwd_B{B}_n{n}_split{split}  {t:9.1f} us   {39 * pairs / t / 1e3:8.1f} Gexp/s", flush=True)
        g1, g2 = torch.empty_like(a), torch.empty_like(b)
        t = timeit(lambda: check(L.l3d_emd_backward(p(a), p(b), p(match), B, n, m, p(g1), p(g2), stream_ptr()), "emd bwd"), warm=2, iters=10)
        print(f"emd_bwd_B{B}_n{n}         {t:9.1f} us   {4.0 * pairs / t / 1e3:8.1f} GB/s (match read once)", flush=True)
        if V1 is not None and B <= 64:
            temp = torch.empty((B, 2 * (n + m)), device="cuda")
            t = timeit(lambda: V1.l3d_emd_forward_v1(p(a), p(b), B, n, m, p(match), p(cost), p(temp), stream_ptr()), warm=1, iters=3, reps=2)
            print(f"emd_fwd_v1_B{B}_n{n}       {t:9.1f} us   (round-4 kernel)", flush=True)


if __name__ == "__main__":
    main()
