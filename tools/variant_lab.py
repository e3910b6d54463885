#!/usr/bin/env python3
"""A/B harness for kernel variants: one translation unit of learning3d_amd/csrc is compiled with extra -D flags into its own
small shared library under tools/bin/ (git-ignored; travels to the GPU box), and its entry point is driven with the SAME real
inputs as the product library's -- outputs compared byte for byte, times interleaved on one box in one process.

    python tools/variant_lab.py build  ef  base=  nostore=-DEF_DIAG_NOSTORE ...      (here: cross-compiles, no GPU)
    python tools/variant_lab.py run    ef  [names...]                                (on the GPU box)
families: ef = edgeconv_f16b.hip / l3d_edgeconv_forward_f16b (out_mode 2, the bench step's launch)
          cf = conv_f16.hip / l3d_pointwise_conv_f16 (conv5 of the bench step, two planes)
Not a product path."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "bin")
CSRC = os.path.join(ROOT, "learning3d_amd", "csrc")
FAM = {"ef": "edgeconv_f16b.hip", "cf": "conv_f16.hip"}


def build(fam, specs):
    from learning3d_amd.build import FLAGS, HIPCC
    os.makedirs(BIN, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        src = FAM[fam]
        fl = flags.split(",") if flags else []
        for f in list(fl):                       # src:<file> swaps the translation unit (an experimental copy under tools/experiments)
            if f.startswith("src:"):
                src = f[4:]
                fl.remove(f)
        srcp = src if os.path.isabs(src) else (os.path.join(ROOT, src) if os.path.exists(os.path.join(ROOT, src)) else os.path.join(CSRC, src))
        out = os.path.join(BIN, f"lib{fam}_{name}.so")
        cmd = [HIPCC, *FLAGS, f"-I{CSRC}", "-shared", *fl, srcp, os.path.join(CSRC, "api.hip"), "-o", out]
        procs.append((name, out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    ok = True
    for name, out, p in procs:
        txt, _ = p.communicate()
        errs = [l for l in txt.splitlines() if "error" in l.lower()]
        print(f"[{fam}:{name}] rc {p.returncode} -> {out}" + ("".join("\n   " + e for e in errs[:8]) if p.returncode else ""))
        ok &= p.returncode == 0
    if ok:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_meta
        for name, out, _ in procs:
            try:
                for kname, k in kernel_meta.kernel_metadata(out).items():
                    if k.get(".private_segment_fixed_size") or k.get(".vgpr_spill_count"):
                        print(f"   !! {name}: {kname[:60]} scratch {k.get('.private_segment_fixed_size')} vspill {k.get('.vgpr_spill_count')}")
            except Exception as exc:
                print(f"   (kernel_meta: {type(exc).__name__}: {exc})")
    return 0 if ok else 1


def timeit(fn, warm, iters, reps=3):
    import torch
    best = None
    for _ in range(reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
        best = t if best is None else min(best, t)
    return best


def run(fam, names):
    import torch
    import learning3d_amd.utils as U
    from learning3d_amd._lib import lib, ptr, stream_ptr
    from learning3d_amd.models import DGCNN, _fused
    names = names or sorted(f[len(fam) + 4:-3] for f in os.listdir(BIN) if f.startswith(f"lib{fam}_") and f.endswith(".so"))
    g = torch.Generator().manual_seed(1000)
    B, N, k = int(os.environ.get("L3D_LAB_B", "32")), 1024, 20
    x = torch.rand((B, N, 3), generator=g).cuda()
    torch.manual_seed(1)
    net = DGCNN(emb_dims=1024).cuda().eval()
    with torch.no_grad():
        idx = U.knn(x.permute(0, 2, 1), k)
        packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        img = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True, unscaled=True)
        w5, s5, b5, w5s, w5f = net._conv5_folded()
        ref5 = _fused.pointwise_conv_f16(img, B, N, w5f, 512, 1024, s5, b5, relu=True, unscaled=True)
        torch.cuda.synchronize()
    flag = _fused.range_flag(x.device)
    rows = []
    if fam == "ef":
        flop = B * N * k * 2 * (6 * 64 + 64 * 64 + 64 * 128 + 128 * 256)
        out = torch.empty_like(img)

        def call(fn):
            rc = fn(ptr(x), ptr(idx), B, N, k, ptr(packed), ptr(out), 2, ptr(flag), stream_ptr())
            assert rc == 0, rc
        variants = [("product", lib().l3d_edgeconv_forward_f16b)]
        ref_name = os.environ.get("L3D_LAB_REF")                   # compare against this variant's output instead of the product's
        for n in names:
            L = ctypes.CDLL(os.path.join(BIN, f"lib{fam}_{n}.so"))
            fn = L.l3d_edgeconv_forward_f16b
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            fn.restype = ctypes.c_int
            variants.append((n, fn))
        if ref_name:
            out.zero_()
            call(dict(variants)[ref_name])
            torch.cuda.synchronize()
            img = out.clone()
        for rnd in range(2):                       # two interleaved rounds: box drift shows up as a difference between them
            for n, fn in variants:
                out.zero_()
                call(fn)
                torch.cuda.synchronize()
                same = bool(torch.equal(out, img))
                nd = int((out != img).sum())
                t = timeit(lambda: call(fn), warm=150 if rnd == 0 else 50, iters=100)
                rows.append((rnd, n, t, flop / t / 1e6, same, nd))
                print(f"round {rnd}  {n:24s} {t:8.1f} us  {flop / t / 1e6:7.1f} TF fp32-equiv  frac {flop / t / 1e6 / 833.3:5.3f}   "
                      f"bytes identical to product: {same} ({nd} differ)", flush=True)
    else:
        flop = B * N * 2 * 512 * 1024
        y = torch.empty_like(ref5)
        s5c, b5c = _fused.f32c(s5), _fused.f32c(b5)

        def call5(fn):
            rc = fn(ptr(img), ptr(w5f), ptr(s5c), ptr(b5c), 0, B, 512, 1024, N, 1, 1, ptr(y), None, None, None, None, 0, None, 0, stream_ptr())
            assert rc == 0, rc
        variants = [("product", lib().l3d_pointwise_conv_f16)]
        for n in names:
            Lb = ctypes.CDLL(os.path.join(BIN, f"lib{fam}_{n}.so"))
            fn = Lb.l3d_pointwise_conv_f16
            fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            fn.restype = ctypes.c_int
            variants.append((n, fn))
        for rnd in range(2):
            for n, fn in variants:
                y.zero_()
                call5(fn)
                torch.cuda.synchronize()
                same = bool(torch.equal(y, ref5))
                nd = int((y != ref5).sum())
                t = timeit(lambda: call5(fn), warm=150 if rnd == 0 else 50, iters=100)
                print(f"round {rnd}  {n:24s} {t:8.1f} us  {flop / t / 1e6:7.1f} TF fp32-equiv  frac {flop / t / 1e6 / 833.3:5.3f}   "
                      f"identical to product: {same} ({nd} differ)", flush=True)
    _fused.check_range(x.device, sync=True)
    return 0


if __name__ == "__main__":
    mode, fam = sys.argv[1], sys.argv[2]
    sys.exit(build(fam, sys.argv[3:]) if mode == "build" else run(fam, sys.argv[3:]))
