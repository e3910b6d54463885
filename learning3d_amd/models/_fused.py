"""Host-side glue for the fp32-MFMA shared-MLP kernels (mlp.hip): BatchNorm folding, parameter
packing (cached per parameter version) and the two launches.  Inference / no-grad only: with
autograd or train-mode BatchNorm (batch statistics) the modules route the 1x1 convs through torch
(rocBLAS/MIOpen) on top of the HIP kNN / grouping kernels -- the training path is ranked under
"next" in SURVEY.md 8(f)."""
import ctypes as C

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


def stage(name):
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


def can_fuse(module, *tensors):
    """Fused MFMA path: inference semantics only (no grad, BatchNorm in eval mode)."""
    if module.training and any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in module.modules()):
        return False
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or
                                    any(p.requires_grad for p in module.parameters())):
        return False
    return True


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
    out = torch.empty(lib().l3d_conv_f16_weight_bytes(Cout, Cin), dtype=torch.uint8, device=w.device)
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
    out = torch.empty(lib().l3d_f16_act_bytes(rows, C), dtype=torch.uint8, device=x.device)
    check(lib().l3d_split_f16_rows(ptr(x), rows, C, int(channel_first), N, ptr(out), ptr(range_flag(x.device)), stream_ptr()),
          "l3d_split_f16_rows")
    return out


def f16_eligible(Cin, Cout, N):
    """conv_f16_kernel's tiles: 256 x 256 (Cout % 256 == 0), else 128 x 512 (Cout % 128 == 0)"""
    return Cin % 16 == 0 and Cin >= 32 and ((Cout % 256 == 0 and N % 256 == 0) or (Cout % 128 == 0 and N % 512 == 0))


def pointwise_conv_f16(x_planes, B, N, w_planes, Cin, Cout, scale=None, shift=None, relu=False, out_planes=False, amax=None):
    """l3d_pointwise_conv_f16 on pre-split operands -> y [B,Cout,N] fp32; out_planes=True: the output as an fp16 activation
    image (uint8 tensor) for the next f16x2 layer instead (l3d_pointwise_conv_f16_planes; shift must be [Cout] or None).
    amax = (int32 tensor, channels per group): also max|y| per channel group as float bits, atomically maximised into the
    (pre-zeroed) tensor (l3d_pointwise_conv_f16_absmax)."""
    scale = f32c(scale) if scale is not None else None
    shift = f32c(shift) if shift is not None else None
    if out_planes:
        if shift is not None and shift.dim() != 1:
            raise ValueError("plane output takes a per-channel shift only")
        dev = x_planes.device
        obs = torch.stack([shift.abs().max() if shift is not None else torch.zeros((), device=dev),
                           scale.abs().max() if scale is not None else torch.ones((), device=dev)]).float()
        img = torch.empty(lib().l3d_f16_act_bytes(B * N, Cout), dtype=torch.uint8, device=dev)
        check(lib().l3d_pointwise_conv_f16_planes(ptr(x_planes), ptr(w_planes), ptr(scale), ptr(shift), ptr(obs), B, Cin, Cout, N,
                                                  int(relu), ptr(img), stream_ptr()), "l3d_pointwise_conv_f16_planes")
        return img
    bstride = Cout if (shift is not None and shift.dim() == 2) else 0
    y = torch.empty((B, Cout, N), dtype=torch.float32, device=x_planes.device)
    if amax is not None:
        check(lib().l3d_pointwise_conv_f16_absmax(ptr(x_planes), ptr(w_planes), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N,
                                                  int(relu), ptr(y), ptr(amax[0]), int(amax[1]), stream_ptr()),
              "l3d_pointwise_conv_f16_absmax")
        return y
    check(lib().l3d_pointwise_conv_f16(ptr(x_planes), ptr(w_planes), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N, int(relu),
                                       ptr(y), stream_ptr()), "l3d_pointwise_conv_f16")
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
    img = torch.empty(lib().l3d_f16_act_bytes(B * N, Cout), dtype=torch.uint8, device=x.device)
    check(lib().l3d_first_layer_f16_planes(ptr(x), int(channel_last), ptr(w), ptr(shift), ptr(xmax), B, Cin, Cout, N, int(relu),
                                           ptr(img), ptr(range_flag(x.device)), stream_ptr()), "l3d_first_layer_f16_planes")
    return img


def pointwise_conv_f16_pool(x_planes, B, N, w_planes, Cin, Cout, scale=None, shift=None, relu=False, out_planes=False, pool=True,
                            group=None):
    """l3d_pointwise_conv_f16_pool: the layer's output as an activation image (out_planes) and / or its maximum over all N points
    [B,Cout] (pool; per-128-point maxima from the kernel's epilogue, then a reduce over N/128) -- or, with group=K (8, 16, 32, 64),
    the maximum over every K consecutive points [B,Cout,N/K] (a grouped layer's max over its K neighbours).
    shift [Cout] or per cloud [B,Cout].  Returns (img or None, pooled or None)."""
    scale = f32c(scale) if scale is not None else None
    shift = f32c(shift) if shift is not None else None
    dev = x_planes.device
    bstride = Cout if (shift is not None and shift.dim() == 2) else 0
    obs = img = part = None
    if out_planes:
        obs = torch.stack([shift.abs().max() if shift is not None else torch.zeros((), device=dev),
                           scale.abs().max() if scale is not None else torch.ones((), device=dev)]).float()
        img = torch.empty(lib().l3d_f16_act_bytes(B * N, Cout), dtype=torch.uint8, device=dev)
    pk = int(group) if group else 128
    if pool or group:
        part = torch.empty((B, Cout, N // pk), dtype=torch.float32, device=dev)
    check(lib().l3d_pointwise_conv_f16_pool(ptr(x_planes), ptr(w_planes), ptr(scale), ptr(shift), bstride, ptr(obs), B, Cin, Cout, N,
                                            int(relu), ptr(img), ptr(part), pk, stream_ptr()), "l3d_pointwise_conv_f16_pool")
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
        check(lib().l3d_pointwise_conv_split_maxpool(ptr(x), int(channel_last), ptr(w_split), ptr(scale), ptr(shift), bstride, B, Cin, Cout,
                                                     N, int(relu), pool, ptr(y), stream_ptr()),
              "l3d_pointwise_conv_split_maxpool")
        return y
    check(lib().l3d_pointwise_conv_maxpool(ptr(x), int(channel_last), ptr(w), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N, int(relu),
                                           pool, ptr(y), stream_ptr()), "l3d_pointwise_conv_maxpool")
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
                                             B, Cin, Cout, N, int(relu), ptr(y), stream_ptr()),
              "l3d_pointwise_conv_split")
        return y
    check(lib().l3d_pointwise_conv(ptr(x), int(channel_last), ptr(w), ptr(scale), ptr(shift), bstride, B, Cin, Cout, N,
                                   int(relu), ptr(y), stream_ptr()), "l3d_pointwise_conv")
    return y


class EdgeConvParams:
    """Folded + fragment-packed parameters of a 4-layer EdgeConv stack, cached on the module and
    rebuilt only when a parameter / running statistic changes (tensor._version) or moves device."""

    def __init__(self):
        self.key = None
        self.packed = None

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
            check(lib().l3d_edgeconv_pack_mag(arr(ws), arr(scs), arr(shs), (C.c_float * 4)(*mags), *cs, ptr(packed)),
                  "l3d_edgeconv_pack_mag")
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
    return GEMM_ARITH if SPLIT_BF16 else "fp32"


class L3DRangeError(RuntimeError):
    pass


_RANGE_FLAGS = {}


def range_flag(device):
    """One int32 in pinned (device-mapped) host memory per GPU.  A f16x2 kernel stores 1 into it when an activation
    leaves fp16's range (never for BatchNorm'd networks); the host reads it without a device sync."""
    key = torch.device(device).index or 0
    f = _RANGE_FLAGS.get(key)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32).pin_memory()
        _RANGE_FLAGS[key] = f
    return f


def check_range(device=None, sync=False):
    """Raise if a f16x2 kernel launched earlier on `device` (default: every device used) reported an out-of-range
    activation: its outputs were invalid.  sync=True waits for the device first (tests, end of a bench run); the
    default looks at what has already completed -- the models call it at the start of every fused forward, so an
    overflow is reported at the next call at the latest."""
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


# EdgeConv kernel choice: "f16" / "split" / "chained" are the register-chained kernels in the three arithmetics above
# (k <= 20); "lds" = the LDS-staged fp32-MFMA kernel (mlp.hip, any k <= 32).  None: by gemm_arith().
EDGECONV_KERNEL = None


def edgeconv_forward(xyz_bn3, idx, packed, widths=(64, 64, 128, 256), kernel=None, planes=False):
    """xyz [B,N,3], idx int64 [B,N,k] -> pooled [B,N,sum(widths)] (channel-last).  planes=True (f16 kernel only): an
    fp16 activation image of the pooled values instead (uint8 tensor), the x operand of pointwise_conv_f16."""
    require_gpu(xyz_bn3, idx, packed)
    B, N, _ = xyz_bn3.shape
    k = idx.shape[2]
    if planes:
        if not (k <= 20 and tuple(widths) == (64, 64, 128, 256)):
            raise ValueError("planes output is produced by the f16 EdgeConv kernel only (k <= 20, 64/64/128/256)")
        out = torch.empty(lib().l3d_f16_act_bytes(B * N, sum(widths)), dtype=torch.uint8, device=xyz_bn3.device)
        check_range(xyz_bn3.device)
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(out), 1, ptr(range_flag(xyz_bn3.device)), stream_ptr())
        with stage("edgeconv_kernel"):                   # the launch alone: a timing span here holds no Python between its
            rc = lib().l3d_edgeconv_forward_f16(*args)   # first event and the kernel (bench.py's live roofline timing)
        check(rc, "l3d_edgeconv_forward_f16")
        return out
    pooled = torch.empty((B, N, sum(widths)), dtype=torch.float32, device=xyz_bn3.device)
    if kernel is None:
        kernel = EDGECONV_KERNEL or {"f16x2": "f16", "bf16x3": "split", "fp32": "chained"}[gemm_arith()]
    if kernel not in ("f16", "split", "chained", "lds"):
        raise ValueError(f"unknown EdgeConv kernel {kernel!r}")
    if kernel != "lds" and (k > 20 or tuple(widths) != (64, 64, 128, 256)):
        kernel = "lds"
    if kernel == "f16":
        check_range(xyz_bn3.device)                      # a previous launch's verdict, if it has completed
        fn, name = lib().l3d_edgeconv_forward_f16, "l3d_edgeconv_forward_f16"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(pooled), 0, ptr(range_flag(xyz_bn3.device)), stream_ptr())
    elif kernel == "split":
        fn, name = lib().l3d_edgeconv_forward_split, "l3d_edgeconv_forward_split"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(pooled), stream_ptr())
    elif kernel == "chained":
        fn, name = lib().l3d_edgeconv_forward_chained, "l3d_edgeconv_forward_chained"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), ptr(pooled), stream_ptr())
    else:
        fn, name = lib().l3d_edgeconv_forward, "l3d_edgeconv_forward"
        args = (ptr(xyz_bn3), ptr(idx), B, N, k, ptr(packed), *widths, ptr(pooled), stream_ptr())
    with stage("edgeconv_kernel"):
        rc = fn(*args)
    check(rc, name)
    return pooled
