"""Multi-GPU layer of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI; "gloo" in the CPU tests), batch sharded contiguously, weights replicated, NO collective on the
data path.  The only exchange is the one the maths needs: ChamferDistanceLoss is a mean over the
WHOLE batch (losses/chamfer_distance.py:38-40), so every rank contributes its partial sums
(sum sqrt d1, sum sqrt d2, point counts) through one all_gather of 4 fp64 values and forms the same
global scalar.  The reference has no distributed layer at all (one nn.DataParallel call,
examples/train_flownet.py:243-245); this replaces its scatter/gather.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  Single-process when WORLD_SIZE is unset or 1 (no process group is created then,
    unless L3D_INIT_SINGLE_RANK=1 asks for a one-rank group: the collectives below then run through RCCL with one rank --
    tests/test_gpu_rccl_one_rank.py, the only way to execute them on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("L3D_INIT_SINGLE_RANK") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(global_batch, rank, world):
    """Contiguous [lo, hi) slice of the batch owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def declare_global_batch(global_batch, world=None):
    """Once per step, when shard_bounds cuts a global batch that the rank count does not divide: tells the train-mode BatchNorm
    exchange every rank's shard size (models/_train.declare_shard_sizes).  None restores the default (equal shards)."""
    from .models import _train
    if global_batch is None:
        return _train.declare_shard_sizes(None)
    world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    spans = [shard_bounds(global_batch, r, world) for r in range(world)]
    _train.declare_shard_sizes([hi - lo for lo, hi in spans])


def shard(tensor, rank, world, declare=False):
    """This rank's contiguous slice of a global batch.  Pure by default (sharding an eval set or a label tensor of
    another length must not change what later BatchNorm exchanges expect).  declare=True also records the sizes it cut
    (declare_global_batch) -- for the training batch of a step, so that the train-mode BatchNorm exchange behind it knows every
    rank's cloud count without asking; the declaration is sticky until the next one (declare_global_batch(None) clears it)."""
    lo, hi = shard_bounds(tensor.shape[0], rank, world)
    if declare and world > 1:
        declare_global_batch(tensor.shape[0], world)
    return tensor[lo:hi]


def combine_chamfer(partials):
    """partials: [world, 4] fp64 rows (sum sqrt d1, sum sqrt d2, n1, n2) -> global loss scalar.
    Device tensors go through the HIP combine kernel; CPU tensors (gloo tests) through torch."""
    if partials.is_cuda:
        from .losses.chamfer_distance import chamfer_combine
        return chamfer_combine(partials)
    tot = partials.sum(dim=0)
    return ((tot[0] / tot[2] + tot[1] / tot[3]) / 2.0).to(torch.float32)


def allgather_chamfer_loss(partial):
    """partial: fp64 tensor [4] = (sum sqrt dist1, sum sqrt dist2, #dist1, #dist2) of this rank's
    shard (device tensor for nccl/RCCL, CPU tensor for gloo).  One all_gather of 32 bytes per rank
    (pure latency over xGMI), then the same combine on every rank: returns the whole-batch Chamfer
    loss, identical everywhere.  No host synchronisation."""
    if dist.is_available() and dist.is_initialized():          # a one-rank group still runs the collective (RCCL with one rank)
        world = dist.get_world_size()
        flat = torch.empty(world * 4, dtype=torch.float64, device=partial.device)
        dist.all_gather_into_tensor(flat, partial.contiguous())
        gathered = flat.view(world, 4)
    else:
        gathered = partial.view(1, 4)
    return combine_chamfer(gathered)


class PipelinedChamferLoss:
    """The same exchange with the collective taken off the critical path: `submit(partial)` starts this
    step's 32-byte all_gather asynchronously (RCCL runs it on its own stream) and returns the loss of
    the PREVIOUS submission, whose gather has had a whole step to complete, so the compute stream never
    waits for xGMI latency or for the slowest rank of the step; `flush()` returns the last one.  Every
    step's loss is still produced, one step late (what asynchronous gradient all-reduce does for
    training).  With one rank there is nothing to hide and `submit` returns the current loss."""

    def __init__(self):
        self._pending = None
        self._multi = dist.is_available() and dist.is_initialized()

    def submit(self, partial):
        if not self._multi:
            return combine_chamfer(partial.view(1, 4))
        world = dist.get_world_size()
        flat = torch.empty(world * 4, dtype=torch.float64, device=partial.device)
        src = partial.contiguous()
        work = dist.all_gather_into_tensor(flat, src, async_op=True)
        prev, self._pending = self._pending, (work, flat, src)
        return self._finish(prev)

    def submit_dists(self, dist1, dist2):
        """submit() from the two distance tensors: with one rank the whole tail (both sqrt-sums + combine) is one
        launch (l3d_chamfer_loss_local); with several, partial sums then the asynchronous exchange."""
        from .losses.chamfer_distance import chamfer_loss_local, chamfer_partials
        if not self._multi and dist1.is_cuda:
            return chamfer_loss_local(dist1, dist2)
        return self.submit(chamfer_partials(dist1, dist2))

    def flush(self):
        prev, self._pending = self._pending, None
        return self._finish(prev)

    @staticmethod
    def _finish(pending):
        if pending is None:
            return None
        work, flat, _src = pending
        work.wait()                      # device tensors: orders the current stream after the collective
        return combine_chamfer(flat.view(-1, 4))


def sharded_chamfer_loss(template_shard, source_shard):
    """ChamferDistanceLoss over a batch that is sharded across ranks (forward / evaluation)."""
    from .losses.chamfer_distance import ChamferDistance, chamfer_partials
    with torch.no_grad():
        d1, d2 = ChamferDistance()(template_shard, source_shard)
        return allgather_chamfer_loss(chamfer_partials(d1, d2))
