"""Two ranks on ONE GPU: the multi-GPU path's HIP kernels executed with world size 2.

A box with a single MI355X cannot host two RCCL ranks (one device per rank), so the 32-byte exchange itself goes through
gloo here (the partial sums are staged through the host for the collective only); everything the ranks COMPUTE is the
product's HIP path: the DGCNN forward and Chamfer search on each rank's shard, l3d_chamfer_partials, and l3d_chamfer_combine
on the gathered partials.  Checked against the single-process whole-batch result: the loss to 1e-6 relative (fp64 partial
sums in a different order), the per-shard features bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from learning3d_amd import parallel
from learning3d_amd.losses.chamfer_distance import ChamferDistance, chamfer_partials
from learning3d_amd.models import DGCNN
rank, world, local = parallel.init_from_env(backend="gloo")
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(0)
x = torch.rand((6, 1024, 3), generator=g); a = torch.rand((6, 512, 3), generator=g); b = torch.rand((6, 700, 3), generator=g)
torch.manual_seed(1)
net = DGCNN(emb_dims=512).cuda().eval()
lo, hi = parallel.shard_bounds(6, rank, world)
with torch.no_grad():
    feat = net(x[lo:hi].cuda())
    d1, d2 = ChamferDistance()(a[lo:hi].cuda(), b[lo:hi].cuda())
    part = chamfer_partials(d1, d2)                                   # device fp64 [4]: this rank's partial sums (HIP)
    flat = torch.empty(world * 4, dtype=torch.float64)
    dist.all_gather_into_tensor(flat, part.cpu())                     # the 32-byte exchange (gloo: one GPU cannot host two RCCL ranks)
    loss = parallel.combine_chamfer(flat.cuda().view(world, 4))       # HIP combine kernel on the gathered partials
torch.cuda.synchronize()
np.savez(os.path.join(sys.argv[2], f"rank{rank}.npz"), feat=feat.cpu().numpy(), loss=float(loss), lo=lo, hi=hi)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_one_gpu_hip_chain(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    # single process, whole batch
    from learning3d_amd.losses import ChamferDistanceLoss
    from learning3d_amd.models import DGCNN
    g = torch.Generator().manual_seed(0)
    x = torch.rand((6, 1024, 3), generator=g); a = torch.rand((6, 512, 3), generator=g); b = torch.rand((6, 700, 3), generator=g)
    torch.manual_seed(1)
    net = DGCNN(emb_dims=512).cuda().eval()
    with torch.no_grad():
        feat = net(x.cuda()).cpu().numpy()
        loss = float(ChamferDistanceLoss()(a.cuda(), b.cuda()))
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["feat"], feat[int(z["lo"]):int(z["hi"])]), f"rank {r}: shard features differ from the whole-batch run"
        assert abs(float(z["loss"]) - loss) <= 1e-6 * abs(loss), (float(z["loss"]), loss)
    assert float(np.load(tmp_path / "rank0.npz")["loss"]) == float(np.load(tmp_path / "rank1.npz")["loss"])


TRAIN_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from learning3d_amd import parallel
from learning3d_amd.models import DGCNN
rank, world, local = parallel.init_from_env(backend="gloo")
torch.cuda.set_device(0)
NB = int(sys.argv[3])
g = torch.Generator().manual_seed(0)
x = torch.rand((NB, 256, 3), generator=g)
w = torch.randn((NB, 128, 256), generator=g)
torch.manual_seed(1)
net = DGCNN(emb_dims=128).cuda().train()
parallel.declare_global_batch(NB)                                 # uneven shards (NB = 5: 3 + 2) are declared, not asked for
lo, hi = parallel.shard_bounds(NB, rank, world)
y = net(x[lo:hi].cuda())                                          # train-mode BatchNorm: per-cloud partials, all_gather, fixed-order sum
(y * w[lo:hi].cuda()).sum().backward()                            # ... and the same for the backward's two batch means
torch.cuda.synchronize()
out = {"y": y.detach().cpu().numpy(), "lo": lo, "hi": hi}
for n, p in net.named_parameters():
    out["g_" + n] = p.grad.cpu().numpy()
for n, b in net.named_buffers():
    out["b_" + n] = b.cpu().numpy()
np.savez(os.path.join(sys.argv[2], f"train{rank}.npz"), **out)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nb", [6, 5])
def test_two_ranks_one_gpu_training_step_statistics(tmp_path, nb):
    """SURVEY.md 8 rows e / f3: a DGCNN training step (train-mode BatchNorm, HIP forward and backward kernels) on two ranks that
    share the GPU, the per-cloud statistic partials exchanged between them (all_gather, staged through the host because the
    group is gloo), against the single-process whole-batch step: outputs, BatchNorm running statistics and the batch means of the
    backward are SHARDING-INVARIANT BY CONSTRUCTION (per-cloud fp64 partials added in global cloud order) -- so each rank's
    outputs and running statistics equal the whole-batch run's bit for bit, even (nb = 6) and uneven (nb = 5) shards alike, and the
    ranks' parameter gradients add up to the whole-batch gradient (1e-5: a different summation order over clouds)."""
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), str(nb)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    from learning3d_amd.models import DGCNN
    g = torch.Generator().manual_seed(0)
    x = torch.rand((nb, 256, 3), generator=g)
    w = torch.randn((nb, 128, 256), generator=g)
    torch.manual_seed(1)
    net = DGCNN(emb_dims=128).cuda().train()
    y = net(x.cuda())
    (y * w.cuda()).sum().backward()
    z = [np.load(tmp_path / f"train{r}.npz") for r in range(2)]
    assert int(z[0]["hi"]) == int(z[1]["lo"]) and int(z[1]["hi"]) == nb
    for r in range(2):
        assert np.array_equal(z[r]["y"], y.detach().cpu().numpy()[int(z[r]["lo"]):int(z[r]["hi"])]), f"rank {r}: outputs differ from the whole-batch step"
        for n, b in net.named_buffers():
            assert np.array_equal(z[r]["b_" + n], b.cpu().numpy()), (r, n)
    for n, p in net.named_parameters():
        want = p.grad.cpu().numpy()
        got = z[0]["g_" + n] + z[1]["g_" + n]
        # every parameter's gradient is the rank's own clouds' share (BatchNorm's dgamma / dbeta included, as with torch's
        # SyncBatchNorm): the data-parallel all_reduce of a training loop adds them up
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 1e-9, (n, float(np.abs(got - want).max()), float(np.abs(want).max()))
