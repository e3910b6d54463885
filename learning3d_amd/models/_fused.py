"""Host-side glue for the shared-MLP kernels: BatchNorm folding, parameter packing (cached per parameter version), the
launches, and the two mechanisms that decide WHICH route a model's forward takes:

  * `checkpointed(module, impl, *tensors)`: the fused kernels are the forward a reference user gets.  The reference's own
    scripts run `model.eval()` with grad mode ON and never call `torch.no_grad()` (examples/test_pointnet.py:31-60,
    examples/test_dcp.py:43-73), so "needs no gradient" cannot be the gate.  Whenever the module's forward is a pure function
    of its inputs and parameters (eval-mode BatchNorm / no BatchNorm, no active dropout) the fused launches run inside an
    autograd.Function; its backward re-evaluates the module through the differentiable per-layer route (_train.py's HIP
    conv / dgrad / wgrad kernels) and back-propagates through that -- activation checkpointing with a faster forward.
  * `run_guarded(device, run)`: f16x2 kernels watch fp16's range; the verdict is read at the end of the SAME call (one event
    wait) and an overflowing call is re-run on bf16x3 (fp32's exponent range) before anything is returned.
Train-mode BatchNorm (batch statistics) takes the per-layer route directly (_train.py)."""
import ctypes as C
import os
import threading

import torch

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr


class _NullSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullSpan()


class StageTimer:
    """Optional per-stage HIP-event timing (bench.py sets `_fused.TIMER = StageTimer()`).  Events are
    recorded on torch's current stream, which is the stream every l3d_* launch is issued on."""

    def __init__(self, only=None):
        self.spans = {}
        self.only = only          # restrict to these span names (None = all)
        self.enabled = True       # flip per step to sample a subset of steps

    def span(self, name):
        if not self.enabled or (self.only is not None and name not in self.only):
            return _NULL
        timer = self

        class _Span:
            def __enter__(self_inner):
                self_inner.e0 = torch.cuda.Event(enable_timing=True)
                self_inner.e1 = torch.cuda.Event(enable_timing=True)
                self_inner.e0.record()

            def __exit__(self_inner, *exc):
                self_inner.e1.record()
                timer.spans.setdefault(name, []).append((self_inner.e0, self_inner.e1))
        return _Span()

    def mean_ms(self):
        """call after torch.cuda.synchronize()"""
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.spans.items()}


TIMER = None
ON_STAGE = None       # optional callable(name), called as a stage is entered and before its launches are issued: lets a caller
                      # fork independent work onto another stream at that point of the chain (bench.py --fork)


def stage(name):
    if ON_STAGE is not None:
        ON_STAGE(name)
    return TIMER.span(name) if TIMER is not None else _NULL


def bn_affine(bn):
    """eval-mode BatchNorm as y = scale * x + shift"""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    shift = bn.bias.detach() - bn.running_mean.detach() * scale
    return scale, shift


def fold_conv_bn(conv, bn=None):
    """conv (1x1 Conv1d/Conv2d or Linear) followed by optional eval-mode BN ->
    (w [Cout,Cin], scale [Cout] or None, shift [Cout] or None) with y = scale*(w x) + shift.
    The result is cached on the conv module per parameter / running-statistic version: folding is five tiny
    elementwise launches per layer, ~0.65 ms per FlowNet3D forward before this cache.  Treat the returned
    tensors as read-only."""
    ts = [conv.weight, conv.bias]
    if bn is not None:
        ts += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)
    hit = conv.__dict__.get("_l3d_fold")
    if hit is not None and hit[0] == key:
        return hit[1]
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).float().contiguous()
    bias = conv.bias.detach().float() if conv.bias is not None else None
    if bn is None:
        out = (w, None, bias)
    else:
        scale, shift = bn_affine(bn)
        if bias is not None:
            shift = shift + scale * bias
        out = (w, scale.float().contiguous(), shift.float().contiguous())
    conv.__dict__["_l3d_fold"] = (key, out)
    return out


def _stochastic_or_batch_dependent(module):
    """train-mode BatchNorm (batch statistics) or an active Dropout: forward is not a pure function of inputs + parameters"""
    if not module.training:
        return False
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training:
            return True
        if isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.training and m.p > 0:
            return True
    return False


def can_fuse(module, *tensors):
    """May this call site launch the fused kernels DIRECTLY: BatchNorm in eval mode and nothing to differentiate.  Inside
    `checkpointed`'s forward grad mode is off, so this is True there; inside its backward recomputation it is False and the
    call site takes its differentiable route."""
    if _stochastic_or_batch_dependent(module):
        return False
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors if t is not None) or
                                    any(p.requires_grad for p in module.parameters())):
        return False
    return True


_TLS = threading.local()


def recomputing():
    return getattr(_TLS, "recomputing", False)


class per_layer_route:
    """`with _fused.per_layer_route():` -- modules called inside with autograd live take their differentiable per-layer route
    directly, exactly as the backward recomputation of `checkpointed` does (tests compare it with the fused forward)."""

    def __enter__(self):
        self.prev = recomputing()
        _TLS.recomputing = True

    def __exit__(self, *exc):
        _TLS.recomputing = self.prev
        return False


def _flatten(out):
    """model output (tensor | tuple / list | dict of tensors, one level) -> (list of tensors, rebuild function)"""
    if isinstance(out, torch.Tensor):
        return [out], lambda ts: ts[0]
    if isinstance(out, dict):
        keys = list(out.keys())
        return [out[k] for k in keys], lambda ts: {k: t for k, t in zip(keys, ts)}
    if isinstance(out, (tuple, list)):
        kind = type(out)
        return list(out), lambda ts: kind(ts)
    raise TypeError(f"checkpointed: unsupported output type {type(out)}")


class _Recompute(torch.autograd.Function):
    """forward: impl(*tensors) with grad mode off (the fused kernels).  backward: impl again with grad mode on, on detached
    inputs (the call sites then take their differentiable routes), and autograd through that graph."""

    @staticmethod
    def forward(ctx, impl, holder, n_in, *args):
        ins = args[:n_in]
        outs, rebuild = _flatten(impl(*ins))
        holder["rebuild"] = rebuild
        ctx.impl, ctx.n_in = impl, n_in
        ctx.save_for_backward(*[a for a in ins if isinstance(a, torch.Tensor)])
        ctx.is_tensor = [isinstance(a, torch.Tensor) for a in ins]
        ctx.non_tensor = [a for a in ins if not isinstance(a, torch.Tensor)]
        ctx.params = list(args[n_in:])          # the module's own Parameter objects (alive as long as the module is)
        # an output that IS an input (identity branches) must be a new tensor object for autograd
        in_ids = {id(a) for a in ins if isinstance(a, torch.Tensor)}
        return tuple(o.view_as(o) if id(o) in in_ids else o for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        saved, others = list(ctx.saved_tensors), list(ctx.non_tensor)
        args = [saved.pop(0) if t else others.pop(0) for t in ctx.is_tensor]
        params = ctx.params
        ins = []
        for i in range(ctx.n_in):
            a = args[i]
            if isinstance(a, torch.Tensor):
                a = a.detach()
                if ctx.needs_input_grad[3 + i]:
                    a.requires_grad_(True)
            ins.append(a)
        from .._lib import on_device_of
        prev = recomputing()
        _TLS.recomputing = True
        try:
            with torch.enable_grad(), on_device_of(*ins):
                outs, _ = _flatten(ctx.impl(*ins))
        finally:
            _TLS.recomputing = prev
        wrt = [a for i, a in enumerate(ins) if isinstance(a, torch.Tensor) and ctx.needs_input_grad[3 + i]]
        wrt += [p for j, p in enumerate(params) if ctx.needs_input_grad[3 + ctx.n_in + j]]
        pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
        got = iter(torch.autograd.grad([o for o, _ in pairs], wrt, [g for _, g in pairs], allow_unused=True) if pairs and wrt
                   else [None] * len(wrt))
        res = [None, None, None]
        for i, a in enumerate(ins):
            res.append(next(got) if isinstance(a, torch.Tensor) and ctx.needs_input_grad[3 + i] else None)
        for j in range(len(params)):
            res.append(next(got) if ctx.needs_input_grad[3 + ctx.n_in + j] else None)
        return tuple(res)


def checkpointed(module, impl, *tensors):
    """Run `impl(*tensors)` (a module's forward body) so that the fused kernels serve the forward whatever the grad mode:
    nothing to differentiate -> impl directly; batch statistics / dropout -> impl directly (its call sites pick the per-layer
    route); otherwise inside _Recompute.  Outputs: a tensor, or a tuple / list / dict of tensors.
    Every route runs under run_guarded: whichever f16x2 kernel a model's forward reaches (PointNet's split rows, the pointer
    network's projections, the SVD head), its range verdict is read inside this call and an overflowing forward is repeated on
    bf16x3 -- no model leaves a raised flag behind for an unrelated later call to trip over."""
    from .._lib import on_device_of
    dev = next((t.device for t in tensors if isinstance(t, torch.Tensor) and t.is_cuda), None)

    def guarded(fn, repeatable=True):
        return run_guarded(dev, fn, repeatable) if dev is not None else fn()

    if not torch.is_grad_enabled() or recomputing() or _stochastic_or_batch_dependent(module):
        with on_device_of(*tensors):                 # tensors on a GPU that is not the current one: switch for the call
            return guarded(lambda: impl(*tensors), repeatable=not _stochastic_or_batch_dependent(module))
    params = [p for p in module.parameters() if p.requires_grad]
    if not params and not any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return guarded(lambda: impl(*tensors))
    if not all(t.is_cuda for t in tensors if isinstance(t, torch.Tensor)):
        return impl(*tensors)
    if TRAIN_DIRECT and module.training and getattr(module, "_l3d_train_direct", False):
        # module.train() with something to learn (examples/train_dcp.py), for the modules that ask for it (the pointer network, the
        # SVD head): a backward WILL follow, so the forward runs once, under autograd, on the differentiable routes -- not once on
        # the fused kernels and again inside the backward (the pointer network's fused forward was 4.5 of a 56 ms DCP step).
        # eval() with grad enabled keeps the fused forward + recompute.
        with on_device_of(*tensors):
            return impl(*tensors)

    def apply():
        holder = {}
        outs = _Recompute.apply(impl, holder, len(tensors), *tensors, *params)
        return holder["rebuild"](list(outs))

    with on_device_of(*tensors):
        return guarded(apply)


# module.train() and something requires grad: run a checkpointed module's forward directly under autograd (see `checkpointed`);
# "0": the fused forward + recompute of rounds 2-5 in every mode (A/B)
TRAIN_DIRECT = os.environ.get("L3D_TRAIN_DIRECT", "1") != "0"

# Training (module.train() with BatchNorm, or autograd through the conv stack): True = conv / dgrad / wgrad on the HIP
# GEMMs with rank-count-independent batch statistics (_train.py); False = torch convs + torch BatchNorm.
TRAIN_HIP = True

# fp32 products as six bf16 MFMA products (conv_split.hip): fp32-equivalent results at 6/16 of the
# fp32-MFMA cost.  False routes every 1x1 conv through the fp32-MFMA kernel of mlp.hip.
SPLIT_BF16 = True


def split_rows(m):
    """fp32 [rows, cols] (device) -> split + tiled bf16x3 image (uint8 tensor) for l3d_pointwise_conv_split"""
    require_gpu(m)
    m = f32c(m)
    rows, cols = m.shape
    out = torch.empty(lib().l3d_split_bytes(rows, cols), dtype=torch.uint8, device=m.device)
    check(lib().l3d_split_rows(ptr(m), rows, cols, ptr(out), stream_ptr()), "l3d_split_rows")
    return out


def split_eligible(Cin, Cout, N):
    return SPLIT_BF16 and Cout % 256 == 0 and N % 128 == 0 and Cin % 16 == 0 and Cin >= 32


def split_weights_f16(w):
    """fp32 [Cout, Cin] (device) -> f16x2 weight image (H | Hs | M planes + 2^-S) for l3d_pointwise_conv_f16"""
    require_gpu(w)
    w = f32c(w)
    Cout, Cin = w.shape
    out = torch.empty(lib().l3d_f16_image_bytes(2, Cout, Cin), dtype=torch.uint8, device=w.device)
    check(lib().l3d_conv_f16_split_weights(ptr(w), Cout, Cin, ptr(out), stream_ptr()), "l3d_conv_f16_split_weights")
    return out


def split_rows_f16(x, channel_first=False):
    """fp32 activations -> activation image (h | m' fp16 planes of x 2^T in the tiled layout of conv_f16.hip, then 2^-T);
    T from the tensor's own maximum.  x [B,N,C] (or [B,C,N] with channel_first).  inf / NaN raise the range flag."""
    require_gpu(x)
    x = f32c(x)
    if channel_first:
        B, C, N = x.shape
    else:
        B, N, C = x.shape
    rows = B * N
    out = torch.empty(lib().l3d_f16_image_bytes(1, rows, C), dtype=torch.uint8, device=x.device)
    check(lib().l3d_split_f16_rows(ptr(x), rows, C, int(channel_first), N, ptr(out), ptr(range_flag(x.device)), stream_ptr()),
          "l3d_split_f16_rows")
    return out


def f16_eligible(Cin, Cout, N):
    """conv_f16_kernel's tiles: 256 x 256 (Cout % 256 == 0), else 128 x 512 (Cout % 128 == 0)"""
    return Cin % 16 == 0 and Cin >= 32 and ((Cout % 256 == 0 and N % 256 == 0) or (Cout % 128 == 0 and N % 512 == 0))


_OBS_CACHE = {}


def _plane_obs(scale, shift, dev):
    """{max|shift|, max|scale|} as a device tensor for the plane-output epilogues; a function of the layer's (folded) parameters
    only, so it is cached per (tensor, version): six tiny launches per layer and call otherwise"""
    key = (str(dev), None if scale is None else (scale.data_ptr(), scale._version), None if shift is None else (shift.data_ptr(), shift._version))
    hit = _OBS_CACHE.get(key)
    if hit is None:
        if len(_OBS_CACHE) > 4096:
            _OBS_CACHE.clear()
        hit = torch.stack([shift.abs().max() if shift is not None else torch.zeros((), device=dev),
                           scale.abs().max() if scale is not None else torch.ones((), device=dev)]).float()
        # keep the keyed tensors alive with the entry: a freed tensor's address could be handed to another parameter
        _OBS_CACHE[key] = (hit, scale, shift)
        return hit
    return hit[0]


def _conv_f16(label, x_planes, w_planes, scale, shift, bstride, B, Cin, Cout, N, relu, flags=0, y=None, residual=None, img=None,
              obs=None, ypool=None, pool=0, amax=None, amax_cdiv=0):
    """the one C entry point of the f16x2 layer (l3d_pointwise_conv_f16); `label` names the variant in the launch log"""
    rc = lib().l3d_pointwise_conv_f16(ptr(x_planes), ptr(w_planes), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N, int(relu),
                                      flags, ptr(y), ptr(residual), ptr(img), ptr(obs), ptr(ypool), int(pool), ptr(amax),
                                      int(amax_cdiv), stream_ptr())
    check(rc, label)


CONV_F16_TWO_PLANE = 1        # include/l3d_hip.h: L3D_CONV_F16_TWO_PLANE
CONV_F16_OUT_UNSCALED = 2     # include/l3d_hip.h: L3D_CONV_F16_OUT_UNSCALED


def pointwise_conv_f16(x_planes, B, N, w_planes, Cin, Cout, scale=None, shift=None, relu=False, out_planes=False, amax=None,
                       unscaled=False, residual=None):
    """l3d_pointwise_conv_f16 on pre-split operands -> y [B,Cout,N] fp32; out_planes=True: the output as an fp16 activation
    image (uint8 tensor) for the next f16x2 layer instead (shift must be [Cout] or None).
    amax = (int32 tensor, channels per group): also max|y| per channel group as float bits, atomically maximised into the
    (pre-zeroed) tensor.  unscaled: the input image carries an unscaled residual plane (the two-plane EdgeConv kernel's, or the
    pointer network's images: l3d_layernorm_planes_cf / l3d_attention_forward_f16b / this layer asked for one) -- the two-plane
    form of the kernel; a plane output then carries an unscaled residual too.
    residual: y = residual + layer(x)."""
    scale = f32c(scale) if scale is not None else None
    shift = f32c(shift) if shift is not None else None
    two = unscaled and Cout % 256 == 0 and N % 256 == 0
    if unscaled and not two:
        raise ValueError("the two-plane conv kernel takes Cout % 256 == 0 and N % 256 == 0")
    if out_planes:
        if shift is not None and shift.dim() != 1:
            raise ValueError("plane output takes a per-channel shift only")
        dev = x_planes.device
        obs = _plane_obs(scale, shift, dev)
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, Cout), dtype=torch.uint8, device=dev)
        _conv_f16("l3d_pointwise_conv_f16[planes]" + ("[two-plane]" if two else ""), x_planes, w_planes, scale, shift, 0, B, Cin, Cout, N,
                  relu, flags=(CONV_F16_TWO_PLANE | CONV_F16_OUT_UNSCALED) if two else 0, img=img, obs=obs)
        return img
    bstride = Cout if (shift is not None and shift.dim() == 2) else 0
    y = torch.empty((B, Cout, N), dtype=torch.float32, device=x_planes.device)
    if residual is not None:
        # y = residual + layer(x): the sublayer's residual connection in the GEMM's epilogue
        if amax is not None or tuple(residual.shape) != (B, Cout, N) or not (Cout % 256 == 0 and N % 256 == 0):
            raise ValueError("residual epilogue: residual [B,Cout,N], Cout % 256 == 0, N % 256 == 0, no absmax")
        _conv_f16("l3d_pointwise_conv_f16[residual]" + ("[two-plane]" if two else ""), x_planes, w_planes, scale, shift, bstride, B, Cin,
                  Cout, N, relu, flags=CONV_F16_TWO_PLANE if two else 0, y=y, residual=f32c(residual))
        return y
    if two and amax is not None:
        _conv_f16("l3d_pointwise_conv_f16[absmax][two-plane]", x_planes, w_planes, scale, shift, bstride, B, Cin, Cout, N, relu,
                  flags=CONV_F16_TWO_PLANE, y=y, amax=amax[0], amax_cdiv=int(amax[1]))
        return y
    if unscaled:
        # the image's residual plane is unscaled (edgeconv_forward(..., planes=True, unscaled=True)): two weight planes
        with stage("conv5_kernel"):
            _conv_f16("l3d_pointwise_conv_f16[two-plane]", x_planes, w_planes, scale, shift, bstride, B, Cin, Cout, N, relu,
                      flags=CONV_F16_TWO_PLANE, y=y)
        return y
    if amax is not None:
        _conv_f16("l3d_pointwise_conv_f16[absmax]", x_planes, w_planes, scale, shift, bstride, B, Cin, Cout, N, relu, y=y,
                  amax=amax[0], amax_cdiv=int(amax[1]))
        return y
    _conv_f16("l3d_pointwise_conv_f16", x_planes, w_planes, scale, shift, bstride, B, Cin, Cout, N, relu, y=y)
    return y


def first_layer_f16_planes(x, w, shift, relu, channel_last):
    """Conv1d(Cin <= 8 -> Cout, k=1) (+ReLU) written straight as the fp16 activation image of the next f16x2 layer
    (l3d_first_layer_f16_planes).  x [B,N,Cin] (channel_last) or [B,Cin,N]."""
    require_gpu(x)
    x, w = f32c(x), f32c(w)
    if channel_last:
        B, N, Cin = x.shape
    else:
        B, Cin, N = x.shape
    Cout = w.shape[0]
    shift = f32c(shift) if shift is not None else None
    xmax = x.abs().max().reshape(1)
    img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, Cout), dtype=torch.uint8, device=x.device)
    check(lib().l3d_first_layer_f16_planes(ptr(x), int(channel_last), ptr(w), ptr(shift), ptr(xmax), B, Cin, Cout, N, int(relu),
                                           ptr(img), ptr(range_flag(x.device)), stream_ptr()), "l3d_first_layer_f16_planes")
    return img


def pointwise_conv_f16_pool(x_planes, B, N, w_planes, Cin, Cout, scale=None, shift=None, relu=False, out_planes=False, pool=True,
                            group=None):
    """l3d_pointwise_conv_f16 with pooled / image outputs: the layer's output as an activation image (out_planes) and / or its maximum over all N points
    [B,Cout] (pool; per-128-point maxima from the kernel's epilogue, then a reduce over N/128) -- or, with group=K (8, 16, 32, 64),
    the maximum over every K consecutive points [B,Cout,N/K] (a grouped layer's max over its K neighbours).
    shift [Cout] or per cloud [B,Cout].  Returns (img or None, pooled or None)."""
    scale = f32c(scale) if scale is not None else None
    shift = f32c(shift) if shift is not None else None
    dev = x_planes.device
    bstride = Cout if (shift is not None and shift.dim() == 2) else 0
    obs = img = part = None
    if out_planes:
        # a per-cloud shift [B,Cout] is data (pcn.py's pooled half of conv3): not cached
        obs = _plane_obs(scale, shift, dev) if (shift is None or shift.dim() == 1) else \
            torch.stack([shift.abs().max(), scale.abs().max() if scale is not None else torch.ones((), device=dev)]).float()
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, Cout), dtype=torch.uint8, device=dev)
    pk = int(group) if group else 128
    if pool or group:
        part = torch.empty((B, Cout, N // pk), dtype=torch.float32, device=dev)
    _conv_f16("l3d_pointwise_conv_f16[pool]", x_planes, w_planes, scale, shift, bstride, B, Cin, Cout, N, relu, img=img, obs=obs,
              ypool=part, pool=pk)
    if group:
        return img, part
    return img, (part.max(dim=2)[0] if pool else None)


def pointwise_conv_maxpool(x, w, scale, shift, relu, pool, w_split=None, channel_last=False):
    """pointwise_conv followed by max over every `pool` consecutive points, in one launch:
    x [B,Cin,S*pool] (or [B,S*pool,Cin] with channel_last) -> [B,Cout,S].  pool in (8, 16, 32, 64); returns None if the
    kernel does not take the shape."""
    require_gpu(x)
    x, w = f32c(x), f32c(w)
    if channel_last:
        B, N, Cin = x.shape
    else:
        B, Cin, N = x.shape
    Cout = w.shape[0]
    if pool not in (8, 16, 32, 64) or N % pool or Cout <= 8:
        return None
    scale = f32c(scale) if scale is not None else None
    shift = f32c(shift) if shift is not None else None
    bstride = Cout if (shift is not None and shift.dim() == 2) else 0
    y = torch.empty((B, Cout, N // pool), dtype=torch.float32, device=x.device)
    if split_eligible(Cin, Cout, N) and N % 256 == 0:
        if w_split is None:
            w_split = split_rows(w)
        check(lib().l3d_pointwise_conv_split(ptr(x), int(channel_last), ptr(w_split), ptr(scale), ptr(shift), bstride, B, Cin, Cout,
                                             N, int(relu), pool, ptr(y), stream_ptr()),
              "l3d_pointwise_conv_split[maxpool]")
        return y
    check(lib().l3d_pointwise_conv(ptr(x), int(channel_last), ptr(w), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N, int(relu),
                                   pool, ptr(y), stream_ptr()), "l3d_pointwise_conv[maxpool]")
    return y


def conv_global_max(x, w, scale, shift, relu):
    """conv (+BN, +ReLU) followed by a max over ALL points: x [B,Cin,N] -> [B,Cout]; the [B,Cout,N] activation
    is not formed when N % 64 == 0 (partial maxima over 64 points in the conv's epilogue, then a tiny reduce)."""
    if x.shape[2] % 64 == 0:
        part = pointwise_conv_maxpool(x, w, scale, shift, relu, 64)
        if part is not None:
            return part.max(dim=2)[0]
    return pointwise_conv(x, w, scale, shift, relu=relu).max(dim=2)[0]


def pointwise_conv(x, w, scale=None, shift=None, relu=False, channel_last=False, w_split=None, split=None):
    """y[b,co,n] = act(scale[co] * sum_ci w[co,ci] x[b,ci,n] + shift[(b,)co]);  x [B,Cin,N] (or
    [B,N,Cin] when channel_last) -> [B,Cout,N].   == Conv1d(k=1) (+BN eval) (+ReLU).
    shift may be [Cout] or per-cloud [B,Cout].  w_split: cached split_rows(w) (else split per call,
    a ~3 us kernel); split=False forces the fp32-MFMA kernel.
    Kernel by shape: bf16x3 (conv_split.hip) when Cout % 256 == 0, N % 128 == 0, Cin % 16 == 0; else the fp32 MFMA (mlp.hip).
    The f16x2 kernel (conv_f16.hip) is NOT routed to from here: it wants its input as fp16 planes, and splitting an fp32
    tensor first (two passes over x: maximum, then split; ~90 us for 67 MB) costs more than the kernel saves for every
    shape but conv5's -- measured on DCP's transformer: 8.15 -> 9.64 ms.  Producers that can emit planes themselves (the
    EdgeConv kernel) call pointwise_conv_f16 directly."""
    require_gpu(x)
    x = f32c(x)
    w = f32c(w)
    if channel_last:
        B, N, Cin = x.shape
    else:
        B, Cin, N = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == Cin
    scale = f32c(scale) if scale is not None else None
    shift = f32c(shift) if shift is not None else None
    bstride = Cout if (shift is not None and shift.dim() == 2) else 0
    y = torch.empty((B, Cout, N), dtype=torch.float32, device=x.device)
    use_split = split_eligible(Cin, Cout, N) if split is None else (split and split_eligible(Cin, Cout, N))
    if use_split:
        if w_split is None:
            w_split = split_rows(w)
        check(lib().l3d_pointwise_conv_split(ptr(x), int(channel_last), ptr(w_split), ptr(scale), ptr(shift), bstride,
                                             B, Cin, Cout, N, int(relu), 0, ptr(y), stream_ptr()),
              "l3d_pointwise_conv_split")
        return y
    check(lib().l3d_pointwise_conv(ptr(x), int(channel_last), ptr(w), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N,
                                   int(relu), 0, ptr(y), stream_ptr()), "l3d_pointwise_conv")
    return y


def linear_rows(x, lin, relu=False):
    """nn.Linear over a handful of rows x [R, Cin] -> [R, Cout] (+ ReLU): l3d_linear_rows (the weight matrix read once); shapes
    it does not take (Cin % 256 != 0) go through the conv kernels with the rows as the points of one cloud"""
    w, _, b = fold_conv_bn(lin)
    x = f32c(x)
    R, Cin = x.shape
    if Cin % 256 == 0:
        y = torch.empty((R, w.shape[0]), dtype=torch.float32, device=x.device)
        check(lib().l3d_linear_rows(ptr(x), ptr(w), ptr(b), R, Cin, w.shape[0], int(relu), ptr(y), stream_ptr()), "l3d_linear_rows")
        return y
    return pointwise_conv(x.unsqueeze(0), w, None, b, relu=relu, channel_last=True)[0].t()


class EdgeConvParams:
    """Folded + fragment-packed parameters of a 4-layer EdgeConv stack, cached on the module and
    rebuilt only when a parameter / running statistic changes (tensor._version) or moves device."""

    def __init__(self):
        self.key = None
        self.packed = None
        self.v2_ok = False

    def get(self, convs, bns, device):
        tensors = []
        for c, b in zip(convs, bns):
            tensors += [c.weight, b.weight, b.bias, b.running_mean, b.running_var]
        key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.key:
            ws, scs, shs, mags = [], [], [], []
            for c, b in zip(convs, bns):
                w, sc, sh = fold_conv_bn(c, b)
                ws.append(w.float().cpu().contiguous())
                scs.append(sc.float().cpu().contiguous())
                shs.append(sh.float().cpu().contiguous())
                # expected post-ReLU magnitude of the layer (4 sigma): the f16x2 kernel places its fp16 planes by it
                mags.append(float((4.0 * b.weight.detach().abs() + b.bias.detach().abs()).max()))
            cs = [w.shape[0] for w in ws]
            nfl = lib().l3d_edgeconv_packed_floats(*cs)
            if nfl == 0:
                raise NotImplementedError(f"EdgeConv channel widths {cs} are not built (64/64/128/256 only)")
            packed = torch.empty(nfl, dtype=torch.float32)
            arr = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
            check(lib().l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), (C.c_float * 4)(*mags), *cs, ptr(packed)),
                  "l3d_edgeconv_pack")
            # the two-plane f16x2 kernel's block is usable when every layer's weights fit its scaling window
            self.v2_ok = bool(packed[lib().l3d_edgeconv_packed_v2_flag_index()] == 1.0)
            self.packed = packed.to(device)
            self.key = key
        return self.packed


# ------------------------------------------------------------------------------------------------------------
# GEMM arithmetic of the shared-MLP kernels (all three are fp32-in / fp32-out with fp32-level error; tests hold the
# matrix-core variants to <= 2x max / 1.5x rms of the fp32-MFMA kernel's own error against fp64):
#   "f16x2"  fp16 high part + 2^12-scaled fp16 residual, 3 fp16 MFMA products per fp32 product (edgeconv_f16.hip);
#            needs |activation| < 65504 -- guarded by a range flag, see range_flag() below
#   "bf16x3" exact three-way bf16 split, 6 bf16 MFMA products per fp32 product (edgeconv_split.hip, conv_split.hip);
#            fp32's full exponent range
#   "fp32"   the fp32 MFMA itself (edgeconv2.hip, mlp.hip)
# SPLIT_BF16 = False (bench.py --fp32-mfma) forces "fp32" regardless.
GEMM_ARITH = "f16x2"


def gemm_arith():
    if not SPLIT_BF16:
        return "fp32"
    return getattr(_TLS, "arith", None) or GEMM_ARITH


class arith:
    """`with _fused.arith("bf16x3"):` -- GEMM arithmetic of the calls inside, for this thread"""

    def __init__(self, name):
        if name not in ("f16x2", "bf16x3", "fp32"):
            raise ValueError(f"unknown GEMM arithmetic {name!r}")
        self.name = name

    def __enter__(self):
        self.prev = getattr(_TLS, "arith", None)
        _TLS.arith = self.name

    def __exit__(self, *exc):
        _TLS.arith = self.prev
        return False


class L3DRangeError(RuntimeError):
    pass


_RANGE_FLAGS = {}


# how often a launch wrapper of THIS THREAD fetched a range flag (run_guarded: no fetch during a call = nothing to wait for) lives in
# _TLS.range_uses, the guard's nesting depth in _TLS.guard_depth: per-GPU threads (DataParallel style) must not see each other's


def range_flag(device):
    """One int32 in pinned (device-mapped) host memory per GPU.  A f16x2 kernel stores 1 into it when an activation
    leaves fp16's range (never for BatchNorm'd networks); the host reads it without a device sync."""
    _TLS.range_uses = getattr(_TLS, "range_uses", 0) + 1     # a kernel is about to be handed the flag: this call can raise it
    key = torch.device(device).index or 0
    f = _RANGE_FLAGS.get(key)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32).pin_memory()
        _RANGE_FLAGS[key] = f
    return f


# What a model does about the f16x2 range flag at the end of its fused forward:
#   "retry"  (default) wait for the call's kernels (one event), read the flag, and if it is set re-run THIS call on bf16x3
#            (fp32's exponent range) -- the caller always receives valid results, like the reference for any input scale;
#   "raise"  the same wait, L3DRangeError instead of the re-run;
#   "async"  no wait: the flag is looked at when the NEXT guarded call starts (check_range) -- maximum launch pipelining for
#            callers that know their activations' range (BatchNorm'd networks on normalised clouds).
# While a hipGraph is being captured nothing can be waited for: the capture owner calls check_range(sync=True) after replays
# (bench.py does).
RANGE_POLICY = "retry"
RANGE_RETRIES = 0          # number of calls re-run on bf16x3 so far (tests, diagnostics)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def range_raised(device, clear=True):
    """Wait for the work queued on the current stream of `device`, then read (and clear) its range flag."""
    f = range_flag(device)
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()
    hit = int(f[0]) != 0
    if hit and clear:
        f[0] = 0
    return hit


def run_guarded(device, run, repeatable=True):
    """run() launches a model's fused forward under the arithmetic gemm_arith() reports and returns its outputs.  With f16x2
    the call's range verdict is read before returning (RANGE_POLICY) and an overflowing call is repeated on bf16x3.  Nested
    calls (DCP -> DGCNN / Transformer / SVDHead, Classifier -> PointNet) run straight through: the outermost one -- every
    model's forward enters through `checkpointed`, which calls this -- waits once and repeats the WHOLE forward if needed.
    The guard is re-entrant per THREAD (depth and flag-use counters are thread-local).  repeatable=False (train-mode BatchNorm,
    dropout: a second run() would update the running statistics and advance the RNG twice for one step): an overflow raises
    L3DRangeError instead of repeating."""
    global RANGE_RETRIES
    if gemm_arith() != "f16x2" or getattr(_TLS, "guard_depth", 0) > 0:
        return run()
    if _capturing() or RANGE_POLICY == "async":
        if not _capturing():
            check_range(device)                      # an earlier call's verdict, if it has completed
        return run()
    _TLS.guard_depth = 1
    try:
        uses = getattr(_TLS, "range_uses", 0)
        out = run()
        if getattr(_TLS, "range_uses", 0) != uses and range_raised(device):      # no f16x2 launch took the flag (FlowNet3D's fp32 stacks): no wait
            if RANGE_POLICY == "raise" or not repeatable:
                raise L3DRangeError("an activation left the fp16 range (|x| > 60000) inside a f16x2 matrix-core kernel"
                                    + (" (_fused.RANGE_POLICY = 'raise')" if repeatable else
                                       " of a forward that cannot be repeated (batch statistics / dropout): run it under "
                                       "_fused.arith('bf16x3')"))
            RANGE_RETRIES += 1
            with arith("bf16x3"):
                out = run()
    finally:
        _TLS.guard_depth = 0
    return out


def check_range(device=None, sync=False):
    """Raise if a f16x2 kernel launched earlier on `device` (default: every device used) reported an out-of-range
    activation: its outputs were invalid.  sync=True waits for the device first (tests, the end of a graph-replay loop);
    the default looks at what has already completed.  Models under RANGE_POLICY "retry" / "raise" never leave a raised
    flag behind; this is the check for "async" callers and for hipGraph replays."""
    keys = list(_RANGE_FLAGS) if device is None else [torch.device(device).index or 0]
    for key in keys:
        f = _RANGE_FLAGS.get(key)
        if f is None:
            continue
        if sync:
            torch.cuda.synchronize(key)
        if int(f[0]) != 0:
            f[0] = 0
            raise L3DRangeError("an activation left the fp16 range (|x| > 60000) inside a f16x2 matrix-core kernel; its "
                                "outputs were invalid.  Set learning3d_amd.models._fused.GEMM_ARITH = 'bf16x3' (full "
                                "fp32 exponent range) and run again.")


# EdgeConv kernel choice: "f16" (f16x2, two weight planes, edgeconv_f16b.hip) and "split" (bf16x3, edgeconv_split.hip) are the
# register-chained kernels (k <= 20, DGCNN's widths); "lds" = the LDS-staged fp32-MFMA kernel (mlp.hip: any k <= 32, any widths,
# plain fp32 arithmetic).  None: by gemm_arith().  One kernel per arithmetic: the three-plane f16 kernel and the register-chained
# fp32 kernel of rounds 1-2 live under tools/experiments/ with the notes on why they were superseded.
EDGECONV_KERNEL = None


def edgeconv_forward(xyz_bn3, idx, packed, widths=(64, 64, 128, 256), kernel=None, planes=False, v2=False, unscaled=False):
    """xyz [B,N,3], idx int64 [B,N,k] -> pooled [B,N,sum(widths)] (channel-last).  planes=True (f16 kernel only): an
    fp16 activation image of the pooled values instead (uint8 tensor), the x operand of pointwise_conv_f16; unscaled: with an
    unscaled residual plane (for the two-plane conv kernel).
    v2: the packed block's two-plane copy is usable (EdgeConvParams.v2_ok) -- what the f16 kernel runs on; without it "f16" falls
    back to the bf16x3 kernel (full fp32 exponent range)."""
    v2 = bool(v2)
    require_gpu(xyz_bn3, idx, packed)
    B, N, _ = xyz_bn3.shape
    k = idx.shape[2]
    if planes:
        if not (v2 and k <= 20 and tuple(widths) == (64, 64, 128, 256)):
            raise ValueError("planes output is produced by the f16 EdgeConv kernel only (usable two-plane block, k <= 20, 64/64/128/256)")
        out = torch.empty(lib().l3d_f16_image_bytes(1, B * N, sum(widths)), dtype=torch.uint8, device=xyz_bn3.device)
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(out), 2 if unscaled else 1, ptr(range_flag(xyz_bn3.device)), stream_ptr())
        with stage("edgeconv_kernel"):                   # the launch alone: a timing span here holds no Python between its
            rc = lib().l3d_edgeconv_forward_f16b(*args)  # first event and the kernel (bench.py's live roofline timing)
        check(rc, "l3d_edgeconv_forward_f16b")
        return out
    if unscaled:
        raise ValueError("an unscaled residual plane belongs to the plane image (planes=True)")
    pooled = torch.empty((B, N, sum(widths)), dtype=torch.float32, device=xyz_bn3.device)
    if kernel is None:
        kernel = EDGECONV_KERNEL or {"f16x2": "f16", "bf16x3": "split", "fp32": "lds"}[gemm_arith()]
    if kernel not in ("f16", "split", "lds"):
        raise ValueError(f"unknown EdgeConv kernel {kernel!r}")
    if kernel != "lds" and (k > 20 or tuple(widths) != (64, 64, 128, 256)):
        kernel = "lds"
    if kernel == "f16" and not v2:
        kernel = "split"
    if kernel == "f16":
        fn, name = lib().l3d_edgeconv_forward_f16b, "l3d_edgeconv_forward_f16b"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(pooled), 0, ptr(range_flag(xyz_bn3.device)), stream_ptr())
    elif kernel == "split":
        fn, name = lib().l3d_edgeconv_forward_split, "l3d_edgeconv_forward_split"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(pooled), stream_ptr())
    else:
        fn, name = lib().l3d_edgeconv_forward, "l3d_edgeconv_forward"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), *widths, ptr(pooled), stream_ptr())
    with stage("edgeconv_kernel"):
        rc = fn(*args)
    check(rc, name)
    return pooled


SA_FUSED = True      # narrow three-layer set-abstraction stacks behind a ball query as ONE kernel (sa_fused.hip); False: group + conv launches


def sa_mlp3_params(convs, bns, dev):
    """The parameter block of l3d_sa_mlp3_fused for three conv + BatchNorm layers (eval mode), cached on the first conv per
    parameter / statistic version; None when the kernel does not take the stack (widths other than 32-32-64 / 64-64-128, more
    than 16 input channels, a layer without BatchNorm scale)."""
    if len(convs) != 3 or len(bns) != 3:
        return None
    folded = [fold_conv_bn(c, b) for c, b in zip(convs, bns)]
    c0 = folded[0][0].shape[1]
    widths = [f[0].shape[0] for f in folded]
    if c0 < 3 or folded[1][0].shape[1] != widths[0] or folded[2][0].shape[1] != widths[1]:
        return None
    if c0 > 16 or tuple(widths) not in ((32, 32, 64), (64, 64, 128)):          # the instantiations of sa_fused.hip
        return None
    # keyed like fold_conv_bn: by the parameters' and running statistics' storage and version (object ids of the folded tensors could
    # be recycled after a re-fold)
    key = tuple((t.data_ptr(), t._version) for m in list(convs) + list(bns) for t in (m.weight, m.bias, getattr(m, "running_mean", None),
                                                                                     getattr(m, "running_var", None)) if t is not None) + (str(dev),)
    hit = convs[0].__dict__.get("_l3d_sa3")
    if hit is not None and hit[0] == key:
        return hit[1]
    parts = []
    c0p = 8 if c0 <= 8 else 16
    for i, (w, sc, sh) in enumerate(folded):
        w = w.to(dev)
        cout, cin = w.shape
        if i == 0 and cin < c0p:
            w = torch.nn.functional.pad(w, (0, c0p - cin))
            cin = c0p
        ns = cin // 4
        run = min(4, ns)
        wl = w.view(cout, ns // run, run, 4).permute(3, 1, 0, 2)                           # [g][s / run][n][s % run] = w[n][4 s + g]
        if run == 2:
            wl = torch.nn.functional.pad(wl, (0, 0, 0, 16))                                  # 16 zero rows per (g, run): bank spread
        parts.append(wl.contiguous().view(-1))
        parts.append((sc if sc is not None else torch.ones(cout, device=dev)).to(dev).float().view(-1))
        parts.append((sh if sh is not None else torch.zeros(cout, device=dev)).to(dev).float().view(-1))
    block = torch.cat(parts).contiguous()
    convs[0].__dict__["_l3d_sa3"] = (key, (block, c0 - 3, widths))
    return block, c0 - 3, widths


def sa_mlp3_fused(xyz_bn3, new_xyz_bs3, feat_bdn, idx, params):
    """max over K of the three-layer shared MLP on [xyz[idx] - new_xyz | feat[idx]] -> [B, C3, S]: one launch, no grouped tensor.
    xyz [B,N,3], new_xyz [B,S,3], feat [B,D,N] or None, idx int32 [B,S,K]; params from sa_mlp3_params."""
    block, D, widths = params
    B, N, _ = xyz_bn3.shape
    S, K = idx.shape[1], idx.shape[2]
    x, q = f32c(xyz_bn3), f32c(new_xyz_bs3)
    f = f32c(feat_bdn) if D > 0 else None
    out = torch.empty((B, widths[2], S), dtype=torch.float32, device=x.device)
    check(lib().l3d_sa_mlp3_fused(ptr(x), ptr(q), ptr(f), ptr(idx.contiguous()), ptr(block), B, N, S, K, D, widths[0], widths[1], widths[2],
                                  ptr(out), stream_ptr()), "l3d_sa_mlp3_fused")
    return out
