"""RCCL executed: the multi-GPU path's collectives on a ONE-rank "nccl" process group.

A one-GPU box cannot host two RCCL ranks, so the N > 1 exchange is otherwise only ever run over gloo (tests/test_gpu_two_ranks.py,
tests/test_host_cpu.py).  With one rank the same product code -- parallel.allgather_chamfer_loss, parallel.PipelinedChamferLoss,
parallel.sharded_chamfer_loss, and the barrier / all_reduce probe bench.py opens a multi-GPU run with -- goes through
ncclCommInitRank, ncclAllGather / ncclAllReduce on device fp64 buffers and the stream ordering of work.wait(): what is checked is
that the RCCL side of the path runs on this image (dmabuf IPC environment included) and returns the single-process numbers."""
import os
import subprocess
import sys

import pytest

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
import torch, torch.distributed as dist
from learning3d_amd import parallel
from learning3d_amd.losses.chamfer_distance import ChamferDistance, ChamferDistanceLoss, chamfer_partials
rank, world, local = parallel.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1, "no one-rank RCCL group"
dev = torch.device("cuda", 0)
probe = torch.full((1,), 1.0, dtype=torch.float64, device=dev)
dist.all_reduce(probe)                                              # bench.py's opening probe
dist.barrier(device_ids=[0])                                        # bench.py's bracket
assert float(probe) == 1.0
g = torch.Generator().manual_seed(0)
a = torch.rand((6, 512, 3), generator=g).to(dev); b = torch.rand((6, 700, 3), generator=g).to(dev)
with torch.no_grad():
    want = float(ChamferDistanceLoss()(a, b))
    d1, d2 = ChamferDistance()(a, b)
    part = chamfer_partials(d1, d2)
    got_sync = float(parallel.allgather_chamfer_loss(part))          # all_gather_into_tensor on device fp64 over RCCL
    pipe = parallel.PipelinedChamferLoss()
    assert pipe._multi
    first = pipe.submit(part)                                        # asynchronous all_gather; nothing pending yet
    second = pipe.submit(part.clone())                               # returns the first submission's loss
    last = pipe.flush()
    got_sharded = float(parallel.sharded_chamfer_loss(a, b))
torch.cuda.synchronize()
assert first is None and abs(float(second) - want) <= 1e-6 * abs(want) and abs(float(last) - want) <= 1e-6 * abs(want), (second, last, want)
assert abs(got_sync - want) <= 1e-6 * abs(want) and abs(got_sharded - want) <= 1e-6 * abs(want), (got_sync, got_sharded, want)
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", got_sync, want)
'''


def test_collectives_run_over_rccl_with_one_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               L3D_INIT_SINGLE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in out, out[-3000:]


@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_bench_multi_rank_path_over_rccl_one_rank(workload):
    """bench.py's N > 1 code path -- RCCL check + all_reduce probe, device barriers around the timed region, the loss exchange inside
    the replayed step (a copy of the graph's partials buffer handed to the asynchronous all_gather), both exchange modes timed,
    per-rank times gathered -- executed on a one-rank nccl group (L3D_INIT_SINGLE_RANK=1)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", L3D_INIT_SINGLE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
           "--workload", workload]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = p.stdout.decode().strip().splitlines()
    assert lines[-1].startswith("{"), lines[-3:]                     # the JSON is the LAST line even with librccl's own stdout chatter
    d = json.loads(lines[-1])
    assert d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl" and d["n_gpus"] == 1, d
    assert d["value"] > 0 and len(d["per_rank_ms_per_step"]) == 1
    if workload == "c2":
        assert d["loss_exchange"].startswith("asynchronous all_gather") and d["other_exchange_mode"]["mode"].startswith("blocking")
        assert d["other_exchange_mode"]["ms_per_step"] > 0
        # the same loss as a plain one-process run of the same seeded step
        q = subprocess.run(cmd[:-2], env={k: v for k, v in os.environ.items()}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert q.returncode == 0, q.stderr.decode()[-3000:]
        e = json.loads(q.stdout.decode().strip().splitlines()[-1])
        assert e["dist_backend"] is None and abs(e["loss"] - d["loss"]) <= 1e-6 * abs(e["loss"]), (e["loss"], d["loss"])
