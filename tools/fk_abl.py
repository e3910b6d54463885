"""featknn ablation timings: python tools/fk_abl.py  (variants tools/bin/libl3d_abl*.so + cap192, built by tools/build_variant_lib.py)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import os, sys, torch
sys.path.insert(0, %r)
import learning3d_amd.utils as U
from tools.kbench import timeit
g = torch.Generator().manual_seed(0)
for B, C, N, k in ((32, 64, 1024, 20), (32, 128, 1024, 20)):
    x = torch.randn((B, C, N), generator=g).cuda()
    print(os.environ.get('L3D_LIB_PATH', 'product')[-16:], B, C, N, k, '%%.1f us' %% timeit(lambda: U.knn(x, k)))
""" % root
for v in sys.argv[1:] or ["cap192", "abl1", "abl2", "abl4", "abl12"]:
    env = dict(os.environ, L3D_LIB_PATH=os.path.join(root, "tools", "bin", f"libl3d_{v}.so"))
    subprocess.run([sys.executable, "-c", code], env=env)
