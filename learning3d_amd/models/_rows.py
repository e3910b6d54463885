"""Differentiable matrix products, row softmax, nn.Linear over rows, square_distance and index_points on the HIP kernels -- the
TRAINING path of the pointer network and the SVD head.  Everything here is l3d_bmm_f32 (strided batched GEMM
on the fp32 matrix cores: an exact fp32 fma chain per output element, no transposed copies) and l3d_softmax_rows, forward and
backward; the forward-only inference path uses the f16x2 kernels instead (utils/transformer.py, utils/svd.py).

reference: utils/transformer.py:127-132 (`attention`: matmul, scale, softmax, matmul), :183-189 / :228-238 (nn.Linear layers),
utils/svd.py:27-31 (scores -> softmax -> src_corr), utils/model_common_utils.py:19-38 (square_distance), :40-56 (index_points);
the gradients are what torch.autograd derives for those op sequences.
"""
import ctypes as C
import os

import torch

from .._lib import check, f32c, lib, on_device_of, ptr, stream_ptr

_ONES = {}


def _ones(n, dev):
    key = (n, str(dev))
    t = _ONES.get(key)
    if t is None:
        if len(_ONES) > 32:
            _ONES.clear()
        t = _ONES[key] = torch.ones(n, dtype=torch.float32, device=dev)
    return t


def _as4(t):
    while t.dim() < 4:
        t = t.unsqueeze(0)
    if t.dim() != 4:
        raise ValueError(f"bmm: at most two batch axes, got shape {tuple(t.shape)}")
    return t


def _st(t):
    return (C.c_long * 4)(*[int(s) for s in t.stride()])


def bmm(a, b, alpha=1.0, out=None, relu=False, bias=None, bias_axis="n", accumulate=False, parts=1):
    """a [..., M, K] @ b [..., K, N] (same number of axes, up to two batch axes, any strides -- transposed, sliced or expanded
    views are read in place) -> act(alpha a b + bias) [..., M, N] fp32.  out: a tensor (any strides) to write / accumulate into."""
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != b.dim() or a.dim() < 2:
        raise ValueError("bmm: fp32 operands with the same number of axes")
    a4, b4 = _as4(a), _as4(b)
    nb1, nb2 = max(a4.shape[0], b4.shape[0]), max(a4.shape[1], b4.shape[1])
    M, K = a4.shape[2:]
    K2, N = b4.shape[2:]
    if K != K2:
        raise RuntimeError(f"bmm: inner dimensions {K} and {K2} differ")
    a4, b4 = a4.expand(nb1, nb2, M, K), b4.expand(nb1, nb2, K, N)
    if out is None:
        out = torch.empty(tuple(torch.broadcast_shapes(a.shape[:-2], b.shape[:-2])) + (M, N), dtype=torch.float32, device=a.device)
    if min(M, N, K, nb1, nb2) == 0:
        return out.zero_() if not accumulate else out
    o4 = _as4(out)
    if tuple(o4.shape) != (nb1, nb2, M, N):
        raise ValueError(f"bmm: out has shape {tuple(out.shape)}, expected batch {nb1} x {nb2} of {M} x {N}")
    ws = None
    if parts > 1:
        ws = torch.empty(nb1 * nb2 * parts * M * N, dtype=torch.float32, device=a.device)
    flags = (1 if accumulate else 0) | (2 if relu else 0) | ((4 if bias_axis == "n" else 8) if bias is not None else 0)
    if bias is not None:
        bias = f32c(bias)
    with on_device_of(a):
        check(lib().l3d_bmm_f32(ptr(a4), _st(a4), ptr(b4), _st(b4), ptr(o4), _st(o4), nb1, nb2, M, N, K, float(alpha), flags, ptr(bias),
                                int(parts), ptr(ws), stream_ptr()), "l3d_bmm_f32")
    return out


def _split_parts(M, N, K):
    """K ranges for a product with few output tiles and a long K (a weight gradient): enough workgroups for the chip, >= 256 of K each"""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    return max(1, min(K // 256, (512 + tiles - 1) // tiles, 64))


class _MatMul(torch.autograd.Function):
    """alpha a @ b for operands with equal batch shapes; da = alpha g b^T, db = alpha a^T g -- read through transposed strides"""

    @staticmethod
    def forward(ctx, a, b, alpha, like):
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        # `like`: a tensor of the product's shape whose memory layout the product takes (the head-split view of a [B, N, h d] tensor:
        # the caller's transpose(1, 2).contiguous() behind the product is then no copy)
        return bmm(a, b, alpha, out=torch.empty_like(like) if like is not None else None)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g if g.dtype == torch.float32 else g.float()
        # a gradient is laid out like its operand (empty_like keeps the strides of a dense view): what autograd does next -- undoing the
        # transposes and head splits the operand came through -- then ends in a contiguous tensor instead of a strided copy
        ga = bmm(g, b.transpose(-1, -2), ctx.alpha, out=_like(a)) if ctx.needs_input_grad[0] else None
        gb = bmm(a.transpose(-1, -2), g, ctx.alpha, out=_like(b)) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None


def _like(t):
    """an uninitialised tensor of t's shape in t's memory layout if t is a dense view (no overlaps, no holes), else contiguous"""
    out = torch.empty_like(t)
    return out if out.stride() == t.stride() else torch.empty(t.shape, dtype=t.dtype, device=t.device)


def matmul(a, b, alpha=1.0, like=None):
    """differentiable alpha * (a @ b) on the HIP GEMM; a [..., M, K], b [..., K, N] with the same batch shape (no broadcasting)"""
    if a.shape[:-2] != b.shape[:-2]:
        raise ValueError("matmul: equal batch shapes (a broadcast operand's gradient would need a reduction)")
    if like is not None and (like.shape != a.shape[:-1] + b.shape[-1:] or like.dtype != torch.float32):
        like = None
    return _MatMul.apply(a.float(), b.float(), float(alpha), like)


class _SoftmaxRows(torch.autograd.Function):
    """p = softmax(scale x) over the last axis; dx = scale p (g - sum_j p_j g_j)"""

    @staticmethod
    def forward(ctx, x, scale):
        xc = f32c(x)
        cols = xc.shape[-1]
        p = torch.empty_like(xc)
        with on_device_of(xc):
            check(lib().l3d_softmax_rows(ptr(xc), None, xc.numel() // cols, cols, float(scale), ptr(p), stream_ptr()), "l3d_softmax_rows")
        ctx.save_for_backward(p)
        ctx.scale = float(scale)
        return p

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = f32c(g)
        cols = p.shape[-1]
        dx = torch.empty_like(p)
        with on_device_of(p):
            check(lib().l3d_softmax_rows(ptr(p), ptr(g), p.numel() // cols, cols, ctx.scale, ptr(dx), stream_ptr()), "l3d_softmax_rows")
        return dx, None


def softmax_rows(x, scale=1.0):
    return _SoftmaxRows.apply(x, float(scale))


def rows_ok(*tensors):
    """the HIP route applies: device fp32 tensors, rows of at most 8192 values for the softmax"""
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def attention_core(query, key, value, scale):
    """softmax(scale q k^T) v and the attention map (reference utils/transformer.py:127-132 without mask / dropout):
    q [..., N, d], k [..., M, d], v [..., M, dv] -> ([..., N, dv], p [..., N, M])"""
    p = softmax_rows(matmul(query, key.transpose(-1, -2)), scale)
    return matmul(p, value, like=query), p


# The Linear layers' forward and dgrad as f16x2 products (three fp16 MFMA products per fp32 product, the arithmetic of the inference path):
# the rows of the batch are the kernel's "weight" operand, the layer's matrix its "activation" operand, so the output is [rows, Cout]
# row-major like everything else here (l3d_split_f16_operand + l3d_pointwise_conv_f16 with TWO_PLANE | SHIFT_N).  One split pass per
# operand (maximum + planes); the weight gradient stays the split-K fp32-MFMA product.  "fp32": every product on l3d_bmm_f32 (rounds 4-5).
TRAIN_GEMM = os.environ.get("L3D_TRAIN_GEMM", "f16x2")
_A_IMG = {}              # the last row operand's image: the q, k, v projections of a self-attention sublayer share their input


def _f16_ok(rows, K, N):
    return TRAIN_GEMM == "f16x2" and rows % 256 == 0 and N % 256 == 0 and K % 16 == 0 and K >= 32 and rows >= 4096


def _operand(t, kind):
    """l3d_split_f16_operand of a 2-D fp32 tensor whose rows are contiguous (row stride >= columns)"""
    rows, Cn = t.shape
    img = torch.empty(lib().l3d_f16_image_bytes(2 if kind else 1, rows, Cn), dtype=torch.uint8, device=t.device)
    check(lib().l3d_split_f16_operand(ptr(t), rows, Cn, t.stride(0), kind, ptr(img), None, stream_ptr()), "l3d_split_f16_operand")
    return img


def _row_operand(x):
    """the rows' image (kind 1), kept for the next call with the SAME tensor (same memory, same version): a self-attention sublayer's
    three projections read one LayerNorm output.  The cache holds the tensor, so its memory cannot be handed to another one meanwhile."""
    key = (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), str(x.device))
    hit = _A_IMG.get("last")
    if hit is not None and hit[0] == key:
        return hit[2]
    img = _operand(x, 1)
    _A_IMG["last"] = (key, x, img)
    return img


def _f16_product(a_img, rows, K, w_img, N, bias, relu):
    """y [rows, N] = act(A B^T + bias): A = the rows' image (kind 1, [rows][K]), B = an activation-kind image of a [N][K] matrix"""
    y = torch.empty((rows, N), dtype=torch.float32, device=a_img.device)
    flags = 1 | (4 if bias is not None else 0)
    check(lib().l3d_pointwise_conv_f16(ptr(w_img), ptr(a_img), None, ptr(bias), 0, 1, K, rows, N, int(bool(relu)), flags,
                                       ptr(y), None, None, None, None, 0, None, 0, stream_ptr()), "l3d_pointwise_conv_f16[rows]")
    return y


class _LinearRows(torch.autograd.Function):
    """y = act(x W^T + b) over rows x [R, Cin]; dx = g W, dW = g^T x (split-K, parts summed in order), db = 1^T g"""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        R, Cin = x.shape
        Cout = w.shape[0]
        if x.is_cuda and x.stride(1) == 1 and x.stride(0) >= Cin and w.is_contiguous() and _f16_ok(R, Cin, Cout):
            with on_device_of(x):
                y = _f16_product(_row_operand(x), R, Cin, _operand(w.detach(), 0), Cout, f32c(b.detach()) if b is not None else None, relu)
        else:
            # W^T as its own [Cin, Cout] tensor (a 1 MB copy): l3d_bmm_f32 then stages it with 16-byte LDS writes; through w.t()'s strides
            # the same product is 15 % slower (scalar LDS writes behind loads along k)
            y = bmm(x, w.t().contiguous() if w.numel() <= (1 << 22) else w.t(), bias=b, bias_axis="n", relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = g if g.dtype == torch.float32 else g.float()
        if ctx.relu:
            g = g * (y > 0)
        R = x.shape[0]
        gx = None
        if ctx.needs_input_grad[0]:
            Cout, Cin = w.shape
            if g.is_cuda and g.dim() == 2 and g.stride(1) == 1 and g.stride(0) >= Cout and _f16_ok(R, Cout, Cin):
                with on_device_of(g):
                    gx = _f16_product(_operand(g, 1), R, Cout, _operand(w.detach().t().contiguous(), 0), Cin, None, False)
            else:
                gx = bmm(g, w)
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = bmm(g.t(), x, parts=_split_parts(w.shape[0], w.shape[1], R))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(g)
        return gx, gw, gb, None


def colsum(g):
    """sum over the rows of g [R, C] (a bias gradient): l3d_colsum_rows, fixed summation order"""
    if g.stride(1) != 1:
        g = g.contiguous()
    R, Cn = g.shape
    out = torch.empty(Cn, dtype=torch.float32, device=g.device)
    with on_device_of(g):
        ws = torch.empty(lib().l3d_colsum_rows_workspace_bytes(R, Cn), dtype=torch.uint8, device=g.device)
        check(lib().l3d_colsum_rows(ptr(g), R, Cn, g.stride(0), ptr(ws), ptr(out), stream_ptr()), "l3d_colsum_rows")
    return out


def linear(x, lin, relu=False):
    """lin(x) (+ ReLU) for x [..., Cin]: rows where they lie (no transposed copies), forward and backward on l3d_bmm_f32"""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    if x2.dtype != torch.float32:
        x2 = x2.float()
    y = _LinearRows.apply(x2, lin.weight, lin.bias, bool(relu))
    return y.view(*shp[:-1], y.shape[1])


class _SquareDistance(torch.autograd.Function):
    """d[b,n,m] = |src_n - dst_m|^2 in the reference's rounding order (l3d_square_distance);
    dsrc = 2 (rowsum(G) src - G dst), ddst = 2 (colsum(G) dst - G^T src)"""

    @staticmethod
    def forward(ctx, src, dst):
        s, d = f32c(src), f32c(dst)
        B, N, Cc = s.shape
        M = d.shape[1]
        out = torch.empty((B, N, M), dtype=torch.float32, device=s.device)
        with on_device_of(s):
            check(lib().l3d_square_distance(ptr(s), ptr(d), B, N, M, Cc, ptr(out), stream_ptr()), "l3d_square_distance")
        ctx.save_for_backward(s, d)
        return out

    @staticmethod
    def backward(ctx, g):
        s, d = ctx.saved_tensors
        g = f32c(g)
        B, N, M = g.shape
        gs = gd = None
        if ctx.needs_input_grad[0]:
            rs = bmm(g, _ones(M, g.device).view(1, M, 1).expand(B, M, 1))                    # [B,N,1]
            gs = bmm(g, d, alpha=-2.0)
            gs.addcmul_(rs, s, value=2.0)
        if ctx.needs_input_grad[1]:
            cs = bmm(g.transpose(1, 2), _ones(N, g.device).view(1, N, 1).expand(B, N, 1))    # [B,M,1]
            gd = bmm(g.transpose(1, 2), s, alpha=-2.0)
            gd.addcmul_(cs, d, value=2.0)
        return gs, gd


def square_distance(src, dst):
    return _SquareDistance.apply(src, dst)


class _IndexPoints(torch.autograd.Function):
    """points[b, idx[b, ...], :] (l3d_index_points); backward: the deterministic scatter-add (l3d_scatter_add_det: entries of
    one target summed in ascending entry order)"""

    @staticmethod
    def forward(ctx, points, idx):
        p = f32c(points)
        B, N, Cc = p.shape
        ix = idx.to(torch.int64).contiguous().view(B, -1)
        S = ix.shape[1]
        out = torch.empty((B, S, Cc), dtype=torch.float32, device=p.device)
        with on_device_of(p):
            check(lib().l3d_index_points(ptr(p), ptr(ix), B, N, Cc, S, ptr(out), stream_ptr()), "l3d_index_points")
        ctx.save_for_backward(ix)
        ctx.n = N
        return out.view(*idx.shape, Cc)

    @staticmethod
    def backward(ctx, g):
        (ix,) = ctx.saved_tensors
        B, S = ix.shape
        Cc = g.shape[-1]
        src = f32c(g.reshape(B, S, Cc).transpose(1, 2))                                      # [B,C,S]
        i32 = ix.to(torch.int32)
        dst = torch.empty((B, Cc, ctx.n), dtype=torch.float32, device=g.device)
        with on_device_of(src):
            ws = torch.empty(lib().l3d_scatter_add_det_workspace_bytes(B, ctx.n, S), dtype=torch.uint8, device=g.device)
            check(lib().l3d_scatter_add_det(ptr(src), ptr(i32), None, B, Cc, ctx.n, S, 1, ptr(ws), ptr(dst), stream_ptr()),
                  "l3d_scatter_add_det")
        return dst.transpose(1, 2), None


def index_points(points, idx):
    return _IndexPoints.apply(points, idx)
