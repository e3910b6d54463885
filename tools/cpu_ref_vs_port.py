"""Calibration of bench.py's cpu_baseline (kind "reference-op-sequence"): the UNMODIFIED reference (imported from a scratch copy
of /root/reference, as tests/golden/make_golden.py does) against oracle.dgcnn_forward_refops + chamfer_loss_refops (what
bench.py times) and against the scalar-C checker port, same inputs, same host, same torch thread count.  Exits non-zero if the
timed restatement runs below 0.9x of the imported reference.
Build container only (needs /root/reference).  usage: python tools/cpu_ref_vs_port.py [clouds] [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
thr = int(sys.argv[2]) if len(sys.argv) > 2 else min(32, os.cpu_count() or 1)
torch.set_num_threads(thr)
import make_golden as MG
tmp, U, Mo, CD = MG.import_reference()
import oracle
g = torch.Generator().manual_seed(0)
x, a, b = (torch.rand((B, 1024, 3), generator=g) for _ in range(3))
torch.manual_seed(1)
net = Mo.DGCNN(emb_dims=1024).eval()
w = {k: v.numpy() for k, v in net.state_dict().items()}


def best(fn, reps=3):
    fn()
    return min((lambda t0: (fn(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(reps))


with torch.no_grad():
    t_ref = best(lambda: (net(x), CD.chamfer(a, b)))
    t_ops = best(lambda: (oracle.dgcnn_forward_refops(x.numpy(), w), oracle.chamfer_loss_refops(a.numpy(), b.numpy())))
    t_port = best(lambda: (oracle.dgcnn_forward_torch(x.numpy(), w), oracle.chamfer_loss(a.numpy(), b.numpy())))
    f_ref, f_port, f_ops = net(x).numpy(), oracle.dgcnn_forward_torch(x.numpy(), w).numpy(), oracle.dgcnn_forward_refops(x.numpy(), w).numpy()
    l_ref, l_ops = float(CD.chamfer(a, b)), float(oracle.chamfer_loss_refops(a.numpy(), b.numpy()))
print(f"host: {os.cpu_count()} logical CPUs, torch threads {thr}, B = {B} clouds x 1024 points, DGCNN(emb 1024).eval() forward + chamfer")
print(f"reference (unmodified, torch fallback Chamfer): {t_ref:.3f} s = {B / t_ref:.1f} clouds/s")
print(f"reference-op-sequence restatement (bench.py cpu_baseline): {t_ops:.3f} s = {B / t_ops:.1f} clouds/s   ratio to reference {t_ref / t_ops:.2f}x")
print(f"oracle port (torch-CPU convs + C kNN / nnsearch): {t_port:.3f} s = {B / t_port:.1f} clouds/s   ratio port/reference speed {t_ref / t_port:.2f}x")
print(f"features: max |reference - port| = {np.abs(f_ref - f_port).max():.2e}")
print(f"features: max |reference - op-sequence restatement| = {np.abs(f_ref - f_ops).max():.2e}; chamfer |diff| = {abs(l_ref - l_ops):.2e}")
if t_ref / t_ops < 0.9 or np.abs(f_ref - f_ops).max() != 0 or l_ref != l_ops:
    sys.exit("cpu_baseline's restatement is slower than 0.9x of the reference, or not bit-identical to it")
