#!/usr/bin/env python3
"""A whole product library with one translation unit rebuilt under extra -D flags, for A/B runs of bench.py / the tests on one box:

    python tools/build_variant_lib.py <name> <unit.hip> [-Dflag ...]      -> tools/bin/libl3d_<name>.so   (git-ignored, travels to the GPU box)
    L3D_LIB_PATH=tools/bin/libl3d_<name>.so python bench.py ...

Every other object comes from learning3d_amd/csrc/build/ (python -m learning3d_amd.build first).  Not a product path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning3d_amd.build import CSRC, FLAGS, HIPCC, build  # noqa: E402


def main():
    name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    build()
    bind = os.path.join(ROOT, "tools", "bin")
    os.makedirs(bind, exist_ok=True)
    objdir = os.path.join(CSRC, "build")
    obj = os.path.join(bind, f"{name}_{unit[:-4]}.o")
    subprocess.check_call([HIPCC, *FLAGS, *flags, "-c", os.path.join(CSRC, unit), "-o", obj])
    objs = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != unit[:-4] + ".o"]
    out = os.path.join(bind, f"libl3d_{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, obj, "-o", out])
    os.remove(obj)
    print(out)


if __name__ == "__main__":
    main()
