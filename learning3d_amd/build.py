"""Build libl3d_hip.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

    python -m learning3d_amd.build          # or: python learning3d_amd/build.py

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the distance kernels replay
the reference's fp32 rounding sequence and write fmaf() explicitly where the reference fuses.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libl3d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "l3d_hip.h")]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        with open(src) as f:                           # variants that #include another kernel source (edgeconv_f16b.hip)
            inc = [os.path.join(CSRC, line.split('"')[1]) for line in f if line.startswith("#include \"") and line.split('"')[1].endswith(".hip")]
        if force or _stale(obj, [src] + inc + hdrs):
            cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
