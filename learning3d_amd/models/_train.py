"""Differentiable shared-MLP layer on the HIP kernels: 1x1 conv / Linear (+ bias) (+ BatchNorm, batch OR running statistics)
(+ ReLU), forward and backward, with batch statistics that are bit-identical for any sharding of the batch across GPUs
(SURVEY.md 8(f) rank 3).  This is the route every model takes whenever autograd is live -- `.train()` with BatchNorm, and
the backward recomputation of `_fused.checkpointed` (eval mode / no BatchNorm with grad enabled).

reference: models/dgcnn.py:34-48 (Conv2d -> BatchNorm2d -> ReLU), models/pointnet.py:22-49, models/pcn.py:26-82, as trained by
examples/train_pcn.py:70-91; the only multi-GPU hook the reference has is nn.DataParallel
(examples/train_flownet.py:243-245), whose BatchNorm statistics are PER-REPLICA.  Here:
  forward   z = W x              HIP GEMM (pointwise_conv: bf16x3 / fp32 MFMA by shape, _fused.py)
            batch statistics:    per-cloud (sum z, sum z^2) fp64              l3d_channel_stats
                                 [all_gather of the partials across ranks]    torch.distributed (RCCL on GPUs, gloo in the CPU test)
                                 added in GLOBAL cloud order -> mean, var,    l3d_bn_finalize (one launch: also the running-statistic
                                 scale / shift, fp64 backward constants        update): same bits whatever the rank count
            running statistics: the layer is the affine map y = scale z + shift (same kernel, mode 1)
            y = relu(z scale + shift)                                         l3d_bn_act_forward
            no BatchNorm: bias and activation in the GEMM's epilogue; y itself is what the backward keeps
  backward  per-cloud (sum g, sum g zhat) fp64                                l3d_bn_backward_stats  [+ all_gather], as SyncBatchNorm
            batch means m1, m2 and dbias / dgamma / dbeta                     l3d_bn_backward_finalize (one launch)
            dz = gr (g - m1 - zhat m2), evaluated in fp64                     l3d_bn_act_backward    (m1 = m2 = 0 without batch statistics)
            dx = W^T dz   (dgrad)                                             HIP GEMM (pointwise_conv with the transposed weight)
            dW = sum_b sum_p dz x^T   (wgrad)                                 l3d_wgrad: split-K on the fp32 MFMA, pieces added in fp64, no atomics
The running statistics follow torch.nn.BatchNorm (momentum update with the unbiased variance, num_batches_tracked).
"""
import torch
import torch.distributed as dist

from .._lib import check, f32c, lib, on_device_of, ptr, stream_ptr
from . import _fused


_SHARD_SIZES = None      # clouds per rank, in rank order, when the shards of the global batch are NOT all equal (declare_shard_sizes)


def declare_shard_sizes(sizes):
    """Tell the BatchNorm statistics exchange how many clouds every rank holds (list in rank order; None: back to discovery).
    parallel.declare_global_batch (and parallel.shard(..., declare=True)) declare the sizes they cut.  The declaration is STICKY
    until replaced or cleared (declare_global_batch(None)); a later step whose local cloud count does not match it raises a
    ValueError on that rank.  Without a declaration the exchange DISCOVERS the sizes on every call (_discover_sizes)."""
    global _SHARD_SIZES
    _SHARD_SIZES = None if sizes is None else [int(v) for v in sizes]


def _discover_sizes(rows, device):
    """Every rank's cloud count through one all_gather of one integer and one host read -- on EVERY undeclared call.  Round 5
    cached the answer per local count; that is wrong exactly where discovery matters: after steps of 4 + 4 clouds, a
    last batch of 4 + 3 lets rank 0 hit its cache for "4" and enter the partials gather while rank 1 (3: unseen) enters this
    exchange -- two different collectives.  No local information can tell a rank that ANOTHER rank's count changed, so there is no
    cache: a training loop that wants the exchange without the host read declares its shards (parallel.declare_global_batch,
    one call per step; equal shards: declare_shard_sizes([n] * world))."""
    world = dist.get_world_size()
    staged = device.type == "cuda" and dist.get_backend() == "gloo"
    mine = torch.tensor([rows], dtype=torch.int64, device="cpu" if staged else device)
    allr = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allr, mine)
    return [int(v) for v in allr.tolist()]


def gather_cloud_partials(part):
    """part [B_local, C, 2] fp64 (this rank's clouds) -> [B_global, C, 2] in global cloud order (ranks hold contiguous
    shards, parallel.shard_bounds).  One collective per call: equal shards go out as they are; unequal shards are padded to
    the largest one and cut back after the gather -- same code on RCCL and gloo.  The per-rank sizes come from the
    declaration (declare_shard_sizes / parallel.declare_global_batch), else from a discovery exchange (one integer per rank and
    a host read, every call) -- undeclared uneven shards never launch a collective with mismatched sizes.
    Single process: returned as is."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world, rank = dist.get_world_size(), dist.get_rank()
        sizes = _SHARD_SIZES
        if sizes is not None and (len(sizes) != world or sizes[rank] != part.shape[0]):
            # a stale declaration cannot be repaired locally (it may still fit the OTHER ranks, who would then enter a different
            # collective): fail loudly on the rank that sees it; torchrun takes the job down
            raise ValueError(f"declared shard sizes {sizes} do not match this rank's {part.shape[0]} clouds (rank {rank} of {world}): "
                             "declare every step (parallel.shard / parallel.declare_global_batch do) or clear it with declare_global_batch(None)")
        if sizes is None:
            sizes = _discover_sizes(int(part.shape[0]), part.device)
        part = part.contiguous()
        # device tensors on a gloo group (two ranks sharing one GPU in tests/test_gpu_two_ranks.py; RCCL wants a device per rank):
        # the few KB go through the host for the collective only
        staged = part.is_cuda and dist.get_backend() == "gloo"

        def gather(src, rows):
            src = src.cpu() if staged else src
            flat = torch.empty((world * rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(flat, src)                       # concatenation along dim 0 in rank order
            return flat.to(part.device) if staged else flat

        if len(set(sizes)) == 1:
            return gather(part, part.shape[0])
        big = max(sizes)
        padded = part if part.shape[0] == big else torch.cat([part, part.new_zeros((big - part.shape[0],) + tuple(part.shape[1:]))])
        flat = gather(padded.contiguous(), big)
        return torch.cat([flat[r * big:r * big + sizes[r]] for r in range(world)], dim=0)
    return part


def sum_clouds(part):
    """[B, ...] fp64 per-cloud partials -> [...] added in cloud order 0, 1, 2, ... (a fixed left-to-right order, independent
    of B's factorisation into ranks); one launch (l3d_sum_clouds_f64).  CPU tensors (the gloo tests): the same order in torch."""
    if not part.is_cuda:
        tot = torch.zeros_like(part[0])
        for b in range(part.shape[0]):
            tot = tot + part[b]
        return tot
    part = part.contiguous()
    tot = torch.empty(part.shape[1:], dtype=torch.float64, device=part.device)
    check(lib().l3d_sum_clouds_f64(ptr(part), part.shape[0], tot.numel(), ptr(tot), stream_ptr()), "l3d_sum_clouds_f64")
    return tot


def stats_from_partials(part_global, points_per_cloud):
    """[B,C,2] fp64 per-cloud sums -> (mean [C], biased var [C], n, totals [C,2]) in fp64"""
    tot = sum_clouds(part_global)
    n = float(part_global.shape[0]) * float(points_per_cloud)
    mean = tot[:, 0] / n
    var = torch.clamp(tot[:, 1] / n - mean * mean, min=0.0)
    return mean, var, n, tot


def channel_stats(z):
    B, C, P = z.shape
    part = torch.empty((B, C, 2), dtype=torch.float64, device=z.device)
    check(lib().l3d_channel_stats(ptr(z), B, C, P, ptr(part), stream_ptr()), "l3d_channel_stats")
    return part


def wgrad(dz, x, pc=0):
    """dW [Cout,Cin] = sum_b dz_b [Cout,P] x_b^T [P,Cin]   (dz [B,Cout,P], x [B,Cin,P] fp32 contiguous): l3d_wgrad"""
    B, Cout, P = dz.shape
    Cin = x.shape[1]
    if pc <= 0:
        # pieces = B * ceil(P / pc): enough workgroups to fill 256 CUs even for a 64 x 6 gradient, a bounded workspace for 1024 x 512
        tiles = ((Cout + 63) // 64) * ((Cin + 63) // 64)
        pc = 2048
        while pc > 256 and tiles * B * ((P + pc - 1) // pc) < 1024:
            pc //= 2
        while B * ((P + pc - 1) // pc) > 65535:
            pc *= 2
    ws = torch.empty(lib().l3d_wgrad_workspace_bytes(B, Cout, Cin, P, pc) // 4, dtype=torch.float32, device=dz.device)
    dw = torch.empty((Cout, Cin), dtype=torch.float32, device=dz.device)
    check(lib().l3d_wgrad(ptr(dz), ptr(x), B, Cout, Cin, P, pc, ptr(ws), ptr(dw), stream_ptr()), "l3d_wgrad")
    return dw


class _ConvAffineAct(torch.autograd.Function):
    """y = act(BN(W x + bias)) for x [B,Cin,P]; bn None: y = act(W x + bias).  batch_stats: BatchNorm with the batch's own
    statistics (train mode), else its running statistics (eval mode: an affine map)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, relu, batch_stats, sync, pool):
        """pool = K > 0: the layer's output is also max-pooled over runs of K consecutive positions (an EdgeConv layer: P = N K, the
        max over the k neighbours); returns (y, ymax [B,Cout,P/K]) and the backward takes both gradients in the two kernels it
        runs anyway -- no dense scatter of the pooled gradient, no add of two [B,Cout,P] tensors."""
        x = f32c(x)                                                      # [B, Cin, P]
        B, Cin, P = x.shape
        w = f32c(weight.reshape(weight.shape[0], -1))
        Cout = w.shape[0]
        dev = x.device
        if bn is None:
            # no BatchNorm: bias and activation ride in the GEMM's epilogue; what is kept for the backward is y itself -- the
            # activation's derivative only needs the sign of the pre-activation, which y has (ReLU and LeakyReLU alike) -- so the
            # layer costs no elementwise forward pass and no second [B,Cout,P] tensor (PCN: 1.5 ms of a 22 ms training step)
            z = _fused.pointwise_conv(x, w, None, f32c(bias.detach()) if bias is not None else None, relu=relu)
            bias = None                                                  # the constants below describe y -> y: scale 1, shift 0
        else:
            z = _fused.pointwise_conv(x, w)                              # HIP GEMM, no epilogue
        n = float(B) * float(P)
        # per-channel constants in ONE launch (l3d_bn_finalize): mean / rstd of the batch (clouds added in cloud order, fp64) or the
        # running statistics or none, the running-statistic update of train mode, gr = gamma rstd, and the fp32 scale / shift
        bias_f = f32c(bias.detach()) if bias is not None else None
        gamma_f = f32c(gamma.detach()) if gamma is not None else None
        beta_f = f32c(beta.detach()) if beta is not None else None
        mean64 = torch.empty(Cout, dtype=torch.float64, device=dev)
        rstd64, gr64 = torch.empty_like(mean64), torch.empty_like(mean64)
        scale = torch.empty(Cout, dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        part_ptr, nb, mode, mom, rm, rv = None, 0, 2, 0.0, None, None
        eps = float(bn.eps) if bn is not None else 0.0
        if bn is not None and batch_stats:
            part = channel_stats(z)
            pg = (gather_cloud_partials(part) if sync else part).contiguous()
            nb, n, mode, part_ptr = pg.shape[0], float(pg.shape[0]) * float(P), 0, ptr(pg)
            if bn.track_running_stats and bn.running_mean is not None:
                if bn.running_mean.dtype != torch.float32 or not bn.running_mean.is_contiguous():
                    raise TypeError("the HIP BatchNorm layer updates fp32 running statistics")
                with torch.no_grad():
                    bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                rm, rv = bn.running_mean, bn.running_var
        elif bn is not None:
            mode, rm, rv = 1, f32c(bn.running_mean.detach()), f32c(bn.running_var.detach())
        check(lib().l3d_bn_finalize(part_ptr, nb, Cout, n, ptr(bias_f), ptr(gamma_f), ptr(beta_f), eps, mode, float(mom), ptr(rm), ptr(rv),
                                    ptr(mean64), ptr(rstd64), ptr(gr64), ptr(scale), ptr(shift), stream_ptr()), "l3d_bn_finalize")
        if bn is None:
            y = z
        else:
            y = torch.empty_like(z)
            check(lib().l3d_bn_act_forward(ptr(z), ptr(scale), ptr(shift), B, Cout, P, int(relu), ptr(y), stream_ptr()), "l3d_bn_act_forward")
        ctx.relu, ctx.sync, ctx.n, ctx.wshape = relu, sync, n, weight.shape
        ctx.batch_stats, ctx.has_bn = bool(bn is not None and batch_stats), bn is not None
        ctx.pool = int(pool)
        if pool:
            if P % pool or pool > 256 or P >= (1 << 22):
                raise ValueError("pooled layer: P must be a multiple of the run length K <= 256 and below 2^22")
            ymax = torch.empty((B, Cout, P // pool), dtype=torch.float32, device=dev)
            pidx = torch.empty(B * Cout * (P // pool), dtype=torch.uint8, device=dev)
            check(lib().l3d_max_last(ptr(y), pidx.numel(), int(pool), ptr(ymax), ptr(pidx), stream_ptr()), "l3d_max_last")
            ctx.save_for_backward(x, w, z, scale, shift, mean64, rstd64, gr64, pidx)
            return y, ymax
        ctx.save_for_backward(x, w, z, scale, shift, mean64, rstd64, gr64)
        return y

    @staticmethod
    def backward(ctx, dy, dmax=None):
        pidx = None
        if ctx.pool:
            x, w, z, scale, shift, mean64, rstd64, gr64, pidx = ctx.saved_tensors
            dmax = f32c(dmax) if dmax is not None else None
            if dmax is None:
                pidx = None
        else:
            x, w, z, scale, shift, mean64, rstd64, gr64 = ctx.saved_tensors
        dy = f32c(dy) if dy is not None else None
        B, Cout, P = z.shape
        if dy is None and dmax is None:
            dy = torch.zeros_like(z)
        K = ctx.pool if pidx is not None else 0
        part = torch.empty((B, Cout, 2), dtype=torch.float64, device=z.device)
        check(lib().l3d_bn_backward_stats(ptr(dy), ptr(z), ptr(scale), ptr(shift), ptr(mean64), ptr(rstd64), B, Cout, P,
                                               int(ctx.relu), ptr(part), ptr(dmax) if pidx is not None else None, ptr(pidx), K,
                                               stream_ptr()), "l3d_bn_backward_stats")
        # (sum g, sum g zhat) over this rank's clouds -> parameter gradients; over every rank's -> the batch means: one launch
        pa = gather_cloud_partials(part).contiguous() if (ctx.batch_stats and ctx.sync) else None
        m1 = torch.empty(Cout, dtype=torch.float64, device=z.device)
        m2 = torch.empty_like(m1)
        dbias = torch.empty(Cout, dtype=torch.float32, device=z.device) if ctx.needs_input_grad[2] else None
        dgamma = torch.empty(Cout, dtype=torch.float32, device=z.device) if (ctx.has_bn and ctx.needs_input_grad[3]) else None
        dbeta = torch.empty(Cout, dtype=torch.float32, device=z.device) if (ctx.has_bn and ctx.needs_input_grad[4]) else None
        check(lib().l3d_bn_backward_finalize(ptr(part), B, ptr(pa), pa.shape[0] if pa is not None else 0, Cout, float(ctx.n),
                                             int(ctx.batch_stats), ptr(gr64), ptr(m1), ptr(m2), ptr(dbias), ptr(dgamma), ptr(dbeta),
                                             stream_ptr()), "l3d_bn_backward_finalize")
        dz = torch.empty_like(z)
        check(lib().l3d_bn_act_backward(ptr(dy), ptr(z), ptr(scale), ptr(shift), ptr(mean64), ptr(rstd64), ptr(gr64), ptr(m1), ptr(m2),
                                             B, Cout, P, int(ctx.relu), ptr(dz), ptr(dmax) if pidx is not None else None, ptr(pidx), K,
                                             stream_ptr()), "l3d_bn_act_backward")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _fused.pointwise_conv(dz, w.t().contiguous())           # dgrad: [B, Cin, P]
        if ctx.needs_input_grad[1]:
            dw = wgrad(dz, x).reshape(ctx.wshape)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None


def conv_bn_act(x, conv, bn=None, relu=True, sync=None):
    """1x1 Conv1d / Conv2d or Linear-shaped module `conv` (weight [Cout,Cin,...], optional bias) -> optional BatchNorm `bn`
    (batch statistics in train mode, running statistics in eval mode) -> optional ReLU, for x [B, Cin, P] or [B, Cin, N, K];
    returns the same rank as x.  sync: share batch statistics across ranks (default: whenever torch.distributed is initialised
    with more than one rank)."""
    if sync is None:
        sync = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    batch_stats = bn is not None and (bn.training or not bn.track_running_stats or bn.running_mean is None)
    shp = x.shape
    x3 = x.reshape(shp[0], shp[1], -1)
    y = _ConvAffineAct.apply(x3, conv.weight, conv.bias, bn.weight if bn is not None else None,
                             bn.bias if bn is not None else None, bn, relu, batch_stats, sync and batch_stats, 0)
    return y.reshape(shp[0], y.shape[1], *shp[2:])


def conv_bn_act_max(x, conv, bn=None, relu=True, sync=None):
    """conv_bn_act for x [B, Cin, N, K] that ALSO returns the max over K (keepdim): (y [B,Cout,N,K], ymax [B,Cout,N,1]) -- an EdgeConv
    layer of models/dgcnn.py:34-46 (`x = relu(bn(conv(x))); x1 = x.max(dim=-1, keepdim=True)[0]`) as ONE autograd node."""
    if sync is None:
        sync = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    batch_stats = bn is not None and (bn.training or not bn.track_running_stats or bn.running_mean is None)
    B, _, N, K = x.shape
    x3 = x.reshape(B, x.shape[1], N * K)
    y, ymax = _ConvAffineAct.apply(x3, conv.weight, conv.bias, bn.weight if bn is not None else None,
                                   bn.bias if bn is not None else None, bn, relu, batch_stats, sync and batch_stats, K)
    return y.reshape(B, y.shape[1], N, K), ymax.unsqueeze(-1)


def linear_act(x, lin, relu=False):
    """nn.Linear over the last axis of x [..., Cin] (+ ReLU) through the same Function: the rows are the points of one cloud."""
    shp = x.shape
    x3 = x.reshape(-1, shp[-1]).t().unsqueeze(0)                          # [1, Cin, rows] (made contiguous by the Function)
    y = _ConvAffineAct.apply(x3, lin.weight, lin.bias, None, None, None, relu, False, False, 0)
    return y[0].t().reshape(*shp[:-1], y.shape[1])


class _MaxLast(torch.autograd.Function):
    """x.max(dim=-1, keepdim=True)[0] for a contiguous fp32 device tensor: l3d_max_last / l3d_max_last_backward"""

    @staticmethod
    def forward(ctx, x):
        K = x.shape[-1]
        R = x.numel() // K
        v = torch.empty(x.shape[:-1] + (1,), dtype=torch.float32, device=x.device)
        idx = torch.empty(R, dtype=torch.uint8, device=x.device)
        with on_device_of(x):
            check(lib().l3d_max_last(ptr(x), R, K, ptr(v), ptr(idx), stream_ptr()), "l3d_max_last")
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return v

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = f32c(g)
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        with on_device_of(g):
            check(lib().l3d_max_last_backward(ptr(g), ptr(idx), idx.numel(), ctx.shape[-1], ptr(gx), stream_ptr()), "l3d_max_last_backward")
        return gx


def max_over_last(x):
    """x.max(dim=-1, keepdim=True)[0] (the max over the k neighbours of an EdgeConv layer, reference models/dgcnn.py:36-46) with a
    one-pass HIP forward and backward where they apply, torch's otherwise"""
    if hip_layers_ok(x) and x.is_contiguous() and x.dim() >= 2 and 1 <= x.shape[-1] <= 256 and x.numel() > 0:
        return _MaxLast.apply(x)
    return x.max(dim=-1, keepdim=True)[0]


class _LayerNormRef(torch.autograd.Function):
    """The pointer network's LayerNorm (reference utils/transformer.py:109-119: unbiased std, eps added to std) over the last
    axis: l3d_layernorm_planes (img NULL) forward, l3d_layernorm_ref_backward (one pass over x and dy; da / db summed in a fixed order)."""

    @staticmethod
    def forward(ctx, x, a, b, eps):
        xc = f32c(x)
        C_ = xc.shape[-1]
        rows = xc.numel() // C_
        y = torch.empty_like(xc)
        ac, bc = f32c(a.detach()), f32c(b.detach())
        with on_device_of(xc):
            check(lib().l3d_layernorm_planes(ptr(xc), ptr(ac), ptr(bc), float(eps), rows, C_, ptr(y), None, stream_ptr()), "l3d_layernorm_planes[values]")
        ctx.save_for_backward(xc, ac)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, ac = ctx.saved_tensors
        g = f32c(g)
        C_ = xc.shape[-1]
        rows = xc.numel() // C_
        dx = torch.empty_like(xc)
        da = torch.empty(C_, dtype=torch.float32, device=xc.device)
        db = torch.empty(C_, dtype=torch.float32, device=xc.device)
        ws = torch.empty(lib().l3d_layernorm_backward_workspace_floats(rows, C_), dtype=torch.float32, device=xc.device)
        with on_device_of(xc):
            check(lib().l3d_layernorm_ref_backward(ptr(xc), ptr(ac), ptr(g), ctx.eps, rows, C_, ptr(dx), ptr(ws), ptr(da), ptr(db),
                                                   stream_ptr()), "l3d_layernorm_ref_backward")
        return dx, da, db, None


def layer_norm_ref(x, a, b, eps):
    """a * (x - mean) / (std + eps) + b over the last axis (unbiased std), differentiable, on the HIP kernels where they apply"""
    C_ = x.shape[-1]
    if hip_layers_ok(x) and C_ % 4 == 0 and 1 < C_ <= 2048 and x.numel() > 0:
        return _LayerNormRef.apply(x, a, b, eps)
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def hip_layers_ok(x):
    """the HIP differentiable layers apply: fp32 tensors on the GPU, switched on (_fused.TRAIN_HIP)"""
    return _fused.TRAIN_HIP and x.is_cuda and x.dtype == torch.float32
