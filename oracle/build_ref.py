"""Compile the reference's own Chamfer C++ CPU path (nnsearch + backward) into
oracle/_ref/cd_ref*.so, from the sources WHERE THEY LIE under /root/reference.

TEST INFRASTRUCTURE ONLY.  Nothing is copied into the repo: the compiler reads
  <ref>/losses/cuda/chamfer_distance/chamfer_distance.cpp
directly; the two CUDA launchers that file declares (defined in the .cu, which
needs nvcc/hipify) are satisfied by a 6-line stub written to oracle/_ref/ so the
module links -- only the CPU entry points `forward` / `backward` are ever called.

The reference's PointNet++ and EMD *extensions* are unbuildable on torch 2.x (THC headers /
AT_CHECK removed), but their KERNELS are plain `__global__` code: build_kernels()
compiles <ref>/utils/lib/src/*_gpu.cu and <ref>/losses/cuda/emd_torch/pkg/include/cuda/emd.cuh for
gfx950 with hipcc (cross-compiles here, no GPU needed) against oracle/ref_compat/ (six CUDA runtime
names -> HIP) plus the C entry files oracle/ref_pointnet2_entry.hip / ref_emd_entry.hip, into
  oracle/_ref/libref_pointnet2.so      -ffp-contract=off  (the arithmetic as written: bit-exact pin)
  oracle/_ref/libref_pointnet2_fma.so  compiler-default contraction (what nvcc's default -fmad does)
  oracle/_ref/libref_chamfer.so        <ref>/losses/cuda/chamfer_distance/chamfer_distance.cu (K1/K2), -ffp-contract=off
  oracle/_ref/libref_emd.so            compiler-default contraction
  oracle/_ref/libref_emd_nofma.so      -ffp-contract=off: the arithmetic as written (the auction amplifies rounding
                                       differences chaotically at n = 1024, so this is the tight pin)
These are the K3-K16 pins of tests/test_gpu_ref_kernels.py on the GPU box.

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


def build_kernels(ref="/root/reference", force=False):
    """hipcc-compile the reference's own PointNet++ / EMD kernels from where they lie."""
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_ref")
    src = os.path.join(ref, "utils", "lib", "src")
    emd_inc = os.path.join(ref, "losses", "cuda", "emd_torch", "pkg", "include")
    cu = [os.path.join(src, f + "_gpu.cu") for f in ("ball_query", "group_points", "interpolate", "sampling")]
    if not all(os.path.exists(c) for c in cu) or not os.path.exists(os.path.join(emd_inc, "cuda", "emd.cuh")):
        print("[oracle/_ref] reference kernel sources not present - skipping (GPU box uses the prebuilt files)")
        return []
    os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-w", f"-I{os.path.join(here, 'ref_compat')}"]
    pn_entry = os.path.join(here, "ref_pointnet2_entry.hip")
    emd_entry = os.path.join(here, "ref_emd_entry.hip")
    cd_entry = os.path.join(here, "ref_chamfer_entry.hip")
    cd_cu = os.path.join(ref, "losses", "cuda", "chamfer_distance", "chamfer_distance.cu")
    jobs = [
        (os.path.join(out, "libref_pointnet2.so"), base + ["-ffp-contract=off", f"-I{src}", *cu, pn_entry], cu + [pn_entry]),
        (os.path.join(out, "libref_pointnet2_fma.so"), base + [f"-I{src}", *cu, pn_entry], cu + [pn_entry]),
        (os.path.join(out, "libref_chamfer.so"), base + ["-ffp-contract=off", cd_cu, cd_entry], [cd_cu, cd_entry]),
        (os.path.join(out, "libref_emd.so"), base + [f"-I{here}", f"-I{emd_inc}", emd_entry],
         [emd_entry, os.path.join(emd_inc, "cuda", "emd.cuh")]),
        (os.path.join(out, "libref_emd_nofma.so"), base + ["-ffp-contract=off", f"-I{here}", f"-I{emd_inc}", emd_entry],
         [emd_entry, os.path.join(emd_inc, "cuda", "emd.cuh")]),
    ]
    built = []
    for so, cmd, deps in jobs:
        if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            print("[oracle/_ref]", " ".join(cmd + ["-o", so]))
            subprocess.check_call(cmd + ["-o", so])
        built.append(so)
    return built


if __name__ == "__main__":
    main(*sys.argv[1:])
    build_kernels(*sys.argv[1:])
