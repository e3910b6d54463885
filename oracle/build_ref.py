"""Compile the reference's own Chamfer C++ CPU path (nnsearch + backward) into
oracle/_ref/cd_ref*.so, from the sources WHERE THEY LIE under /root/reference.

TEST INFRASTRUCTURE ONLY.  Nothing is copied into the repo: the compiler reads
  <ref>/losses/cuda/chamfer_distance/chamfer_distance.cpp
directly; the two CUDA launchers that file declares (defined in the .cu, which
needs nvcc/hipify) are satisfied by a 6-line stub written to oracle/_ref/ so the
module links -- only the CPU entry points `forward` / `backward` are ever called.
The reference's PointNet++ and EMD extensions are unbuildable on torch 2.x
(THC headers / AT_CHECK removed) and are restated in oracle.c.

usage: python build_ref.py [/root/reference]
"""
import os
import subprocess
import sys
import sysconfig


def main(ref="/root/reference"):
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_ref")
    src = os.path.join(ref, "losses", "cuda", "chamfer_distance", "chamfer_distance.cpp")
    if not os.path.exists(src):
        print(f"[oracle/_ref] {src} not present - skipping (GPU box uses the prebuilt file)")
        return None
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "cd_ref" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    stub = os.path.join(out, "cuda_launcher_stub.cpp")
    with open(stub, "w") as f:
        f.write("// link-time stand-ins for the reference's .cu launchers (never called)\n"
                "int ChamferDistanceKernelLauncher(const int, const int, const float*, const int,\n"
                "    const float*, float*, int*, float*, int*) { return -1; }\n"
                "int ChamferDistanceGradKernelLauncher(const int, const int, const float*, const int,\n"
                "    const float*, const float*, const int*, const float*, const int*, float*, float*) { return -1; }\n")
    from torch.utils import cpp_extension as ce
    import torch
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w", "-DTORCH_EXTENSION_NAME=cd_ref",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(torch._C._GLIBCXX_USE_CXX11_ABI)),
           *inc, src, stub, "-o", so, f"-L{libdir}", f"-Wl,-rpath,{libdir}",
           "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    print("[oracle/_ref]", " ".join(cmd))
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    main(*sys.argv[1:])
