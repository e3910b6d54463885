"""Training path of the shared MLP: 1x1 conv + train-mode BatchNorm (+ ReLU) on the HIP kernels, with batch
statistics that are bit-identical for any sharding of the batch across GPUs.

reference: models/dgcnn.py:34-48 (Conv2d -> BatchNorm2d -> ReLU) and models/pcn.py in .train(); the only multi-GPU
hook the reference has is nn.DataParallel (examples/train_flownet.py:243-245), whose BatchNorm statistics are
PER-REPLICA.  Here:
  forward   z = W x              HIP GEMM (pointwise_conv: f16x2 / bf16x3 / fp32 MFMA by shape, _fused.py)
            per-cloud (sum z, sum z^2) fp64            l3d_channel_stats
            [all_gather of the partials across ranks]  torch.distributed (RCCL on GPUs, gloo in the CPU test)
            summed in GLOBAL cloud order -> mean, var  same bits whatever the number of ranks
            y = relu(z scale + shift)                  l3d_bn_act_forward
  backward  per-cloud (sum g, sum g zhat) fp64         l3d_bn_backward_stats  [+ all_gather], as torch's SyncBatchNorm
            dz                                          l3d_bn_act_backward
            dx = W^T dz   (dgrad)                       HIP GEMM (pointwise_conv with the transposed weight)
            dW = sum_b dz_b x_b^T   (wgrad)             HIP GEMM per cloud (points are the K axis), summed in cloud order
The running statistics follow torch.nn.BatchNorm (momentum update with the unbiased variance, num_batches_tracked).
"""
import torch
import torch.distributed as dist

from .._lib import check, f32c, lib, ptr, stream_ptr
from . import _fused


def gather_cloud_partials(part):
    """part [B_local, C, 2] fp64 (this rank's clouds) -> [B_global, C, 2] in global cloud order (ranks hold contiguous
    shards of equal size, parallel.shard_bounds).  Single process: returned as is."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        flat = torch.empty((world * part.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(flat, part.contiguous())             # concatenation along dim 0 in rank order
        return flat
    return part


def stats_from_partials(part_global, points_per_cloud):
    """[B,C,2] fp64 per-cloud sums -> (mean [C], biased var [C], n) in fp64, added in cloud order 0, 1, 2, ..."""
    tot = torch.zeros_like(part_global[0])
    for b in range(part_global.shape[0]):                    # a fixed left-to-right order, independent of B's factorisation
        tot = tot + part_global[b]
    n = float(part_global.shape[0]) * float(points_per_cloud)
    mean = tot[:, 0] / n
    var = torch.clamp(tot[:, 1] / n - mean * mean, min=0.0)
    return mean, var, n, tot


def channel_stats(z):
    B, C, P = z.shape
    part = torch.empty((B, C, 2), dtype=torch.float64, device=z.device)
    check(lib().l3d_channel_stats(ptr(z), B, C, P, ptr(part), stream_ptr()), "l3d_channel_stats")
    return part


class _ConvBNActTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, relu, sync):
        x = f32c(x)                                                      # [B, Cin, P]
        B, Cin, P = x.shape
        w = f32c(weight.reshape(weight.shape[0], -1))
        Cout = w.shape[0]
        z = _fused.pointwise_conv(x, w)                                  # HIP GEMM, no epilogue
        part = channel_stats(z)
        pg = gather_cloud_partials(part) if sync else part
        mean64, var64, n, _ = stats_from_partials(pg, P)
        with torch.no_grad():                                            # running statistics, as torch.nn.BatchNorm in train mode
            if bn.track_running_stats and bn.running_mean is not None:
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                unbiased = var64 * (n / max(n - 1.0, 1.0))
                bn.running_mean.mul_(1 - m).add_(mean64.to(bn.running_mean.dtype), alpha=m)
                bn.running_var.mul_(1 - m).add_(unbiased.to(bn.running_var.dtype), alpha=m)
        rstd64 = torch.rsqrt(var64 + bn.eps)
        scale = (gamma.detach().double() * rstd64).float().contiguous()
        shift = (beta.detach().double() - mean64 * gamma.detach().double() * rstd64).float().contiguous()
        y = torch.empty_like(z)
        check(lib().l3d_bn_act_forward(ptr(z), ptr(scale), ptr(shift), B, Cout, P, int(relu), ptr(y), stream_ptr()), "l3d_bn_act_forward")
        ctx.save_for_backward(x, w, z, scale, shift, mean64.float().contiguous(), rstd64.float().contiguous(), gamma.detach().float().contiguous())
        ctx.relu, ctx.sync, ctx.n, ctx.wshape = relu, sync, n, weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, z, scale, shift, mean, rstd, gamma = ctx.saved_tensors
        dy = f32c(dy)
        B, Cout, P = z.shape
        Cin = x.shape[1]
        part = torch.empty((B, Cout, 2), dtype=torch.float64, device=z.device)
        check(lib().l3d_bn_backward_stats(ptr(dy), ptr(z), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), B, Cout, P, int(ctx.relu), ptr(part),
                                          stream_ptr()), "l3d_bn_backward_stats")
        local = stats_from_partials(part, P)[3]                          # this rank's (sum g, sum g zhat): the parameter gradients
        tot = stats_from_partials(gather_cloud_partials(part), P)[3] if ctx.sync else local
        dbeta, dgamma = local[:, 0].float(), local[:, 1].float()
        gr = (gamma * rstd).contiguous()
        m1, m2 = (tot[:, 0] / ctx.n).float().contiguous(), (tot[:, 1] / ctx.n).float().contiguous()
        dz = torch.empty_like(z)
        check(lib().l3d_bn_act_backward(ptr(dy), ptr(z), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), ptr(gr), ptr(m1), ptr(m2), B, Cout, P,
                                        int(ctx.relu), ptr(dz), stream_ptr()), "l3d_bn_act_backward")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _fused.pointwise_conv(dz, w.t().contiguous())           # dgrad: [B, Cin, P]
        if ctx.needs_input_grad[1]:
            # wgrad: dW = sum_b dz_b [Cout, P] x_b^T [P, Cin]; per cloud the points are the GEMM's K axis: x_b is a
            # "channel-last" [N' = Cin, K' = P] operand and dz_b the [Cout, K'] weight of the same kernel
            acc = torch.zeros((Cout, Cin), dtype=torch.float32, device=z.device)
            for b in range(B):
                acc = acc + _fused.pointwise_conv(x[b:b + 1], dz[b], channel_last=True, split=False)[0]
            dw = acc.reshape(ctx.wshape)
        return dx, dw, (dgamma if ctx.needs_input_grad[2] else None), (dbeta if ctx.needs_input_grad[3] else None), None, None, None


def conv_bn_act(x, conv, bn, relu=True, sync=None):
    """Conv(1x1, no bias or bias folded into BN: a bias before BatchNorm cancels) -> BatchNorm (batch statistics) -> ReLU
    for x [B, Cin, P] or [B, Cin, N, K]; returns the same rank as x.  sync: share the statistics across ranks
    (default: whenever torch.distributed is initialised with more than one rank)."""
    if sync is None:
        sync = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    shp = x.shape
    x3 = x.reshape(shp[0], shp[1], -1)
    y = _ConvBNActTrain.apply(x3, conv.weight, bn.weight, bn.bias, bn, relu, sync)
    return y.reshape(shp[0], y.shape[1], *shp[2:])
