"""Drop-in for learning3d/utils/svd.py on MI355X.

reference: utils/svd.py:5-59.  Two HIP launches: the score GEMM + softmax + weighted target sum as one
flash-style pass that never materialises the [B,N,N] scores (l3d_soft_correspondence, SURVEY.md 8(f)
rank 1), and everything after `src_corr` (centring, H, the per-item torch.svd / det / reflect loop with
its B host syncs, and t) as one kernel (l3d_kabsch).  With autograd enabled the score part runs through
torch ops (the fused kernel is forward-only) and Kabsch through _KabschFunction: HIP forward, analytic backward
(kabsch_backward) -- R and t carry gradients to the embeddings like the reference's torch.svd path does.
"""
import math

import torch
import torch.nn as nn

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr


def kabsch(src, src_corr):
    """src, src_corr [B,3,N] -> R [B,3,3], t [B,3]   (utils/svd.py:29-58)."""
    require_gpu(src, src_corr)
    s, c = f32c(src), f32c(src_corr)
    B, _, N = s.shape
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=s.device)
    t = torch.empty((B, 3), dtype=torch.float32, device=s.device)
    check(lib().l3d_kabsch(ptr(s), ptr(c), B, N, ptr(R), ptr(t), None, stream_ptr()), "l3d_kabsch")
    return R, t


def kabsch_backward(src, src_corr, R, grad_R, grad_t):
    """Adjoint of (R, t) = kabsch(src, src_corr) -- what autograd derives for the reference's sequence
    utils/svd.py:29-58 (centring, H = S C^T, torch.svd, R = V U^T with the reflection fix, t = -R mean(src) +
    mean(corr)), written without the SVD: R is the orthogonal polar factor of H^T = R P, P = R^T H^T symmetric, so
    dR = R [x]_x with (tr(P) I - P) x = vee(M - M^T), M = R^T dH^T.  Hence dL/dH = -[y]_x R^T with
    (tr(P) I - P) y = vee(Q - Q^T), Q = R^T dL/dR.  Singular exactly where torch.svd's backward is (s_i + s_j = 0).
    Pure torch on [B,3,3] / [B,3,N] tensors (device-agnostic; tests check it against autograd on the CPU)."""
    N = src.shape[2]
    mu_s, mu_c = src.mean(dim=2, keepdim=True), src_corr.mean(dim=2, keepdim=True)
    sc, cc = src - mu_s, src_corr - mu_c
    H = torch.matmul(sc, cc.transpose(2, 1))
    gR = grad_R if grad_R is not None else torch.zeros_like(R)
    g_mu_s = g_mu_c = None
    if grad_t is not None:
        gt = grad_t.reshape(-1, 3, 1)
        gR = gR - torch.matmul(gt, mu_s.transpose(2, 1))           # t = -R mu_s + mu_c
        g_mu_s = -torch.matmul(R.transpose(2, 1), gt)
        g_mu_c = gt
    Rt = R.transpose(2, 1)
    P = torch.matmul(Rt, H.transpose(2, 1))
    eye = torch.eye(3, dtype=R.dtype, device=R.device).expand_as(P)
    K = P.diagonal(dim1=1, dim2=2).sum(-1)[:, None, None] * eye - P
    Q = torch.matmul(Rt, gR)
    qv = torch.stack([Q[:, 2, 1] - Q[:, 1, 2], Q[:, 0, 2] - Q[:, 2, 0], Q[:, 1, 0] - Q[:, 0, 1]], dim=1)
    y = torch.linalg.solve(K, qv.unsqueeze(-1)).squeeze(-1)
    z = torch.zeros_like(y[:, 0])
    Yx = torch.stack([torch.stack([z, -y[:, 2], y[:, 1]], 1), torch.stack([y[:, 2], z, -y[:, 0]], 1),
                      torch.stack([-y[:, 1], y[:, 0], z], 1)], 1)
    gH = -torch.matmul(Yx, Rt)
    g_sc = torch.matmul(gH, cc)
    g_cc = torch.matmul(gH.transpose(2, 1), sc)
    g_src = g_sc - g_sc.mean(dim=2, keepdim=True)
    g_corr = g_cc - g_cc.mean(dim=2, keepdim=True)
    if g_mu_s is not None:
        g_src = g_src + g_mu_s / N
        g_corr = g_corr + g_mu_c / N
    return g_src, g_corr


class _KabschFunction(torch.autograd.Function):
    """Differentiable Kabsch: forward = l3d_kabsch (HIP), backward = kabsch_backward.  The reference's head is
    differentiable through torch.svd and DCP's training depends on it (est_R / est_t carry the loss)."""

    @staticmethod
    def forward(ctx, src, src_corr):
        R, t = kabsch(src, src_corr)
        ctx.save_for_backward(src, src_corr, R)
        return R, t

    @staticmethod
    def backward(ctx, grad_R, grad_t):
        src, src_corr, R = ctx.saved_tensors
        g_src, g_corr = kabsch_backward(f32c(src), f32c(src_corr), R, grad_R, grad_t)
        return (g_src if ctx.needs_input_grad[0] else None), (g_corr if ctx.needs_input_grad[1] else None)


def soft_correspondence(src_embedding, tgt_embedding, tgt, scale=None):
    """src_emb [B,C,N], tgt_emb [B,C,M], tgt [B,3,M] -> src_corr [B,3,N]   (utils/svd.py:22-27);
    scale defaults to 1/sqrt(C)."""
    require_gpu(src_embedding, tgt_embedding, tgt)
    q, k, v = f32c(src_embedding), f32c(tgt_embedding), f32c(tgt)
    B, C, N = q.shape
    M = k.shape[2]
    assert k.shape[1] == C and v.shape == (B, 3, M)
    if scale is None:
        scale = 1.0 / math.sqrt(C)
    ws = torch.empty(lib().l3d_soft_correspondence_workspace_floats(B, N, M), dtype=torch.float32, device=q.device)
    out = torch.empty((B, 3, N), dtype=torch.float32, device=q.device)
    check(lib().l3d_soft_correspondence(ptr(q), ptr(k), ptr(v), B, C, N, M, float(scale), ptr(ws), ptr(out),
                                        stream_ptr()), "l3d_soft_correspondence")
    return out


def svd3x3_rotation(H):
    """H [B,3,3] -> R = V U^T with the det < 0 reflection fix   (utils/svd.py:38-49)."""
    require_gpu(H)
    h = f32c(H)
    R = torch.empty_like(h)
    check(lib().l3d_svd3x3_rotation(ptr(h), h.shape[0], ptr(R), stream_ptr()), "l3d_svd3x3_rotation")
    return R


class SVDHead(nn.Module):
    _l3d_train_direct = True       # _fused.checkpointed: in train() mode the forward runs once, on the differentiable routes

    def __init__(self, emb_dims, input_shape="bnc"):
        super(SVDHead, self).__init__()
        self.emb_dims = emb_dims
        # kept so reference checkpoints / state_dicts load unchanged (utils/svd.py:9-10)
        self.reflect = nn.Parameter(torch.eye(3), requires_grad=False)
        self.reflect[2, 2] = -1
        self.input_shape = input_shape

    def forward(self, *input):
        """Flash-style soft correspondences + Kabsch kernel in every grad mode; a backward recomputes the score matrix route
        below and the analytic Kabsch backward (_fused.checkpointed)."""
        from ..models import _fused
        return _fused.checkpointed(self, self._forward, input[0], input[1], input[2], input[3])

    def _forward(self, *input):
        src_embedding, tgt_embedding, src, tgt = input[0], input[1], input[2], input[3]
        batch_size = src.size(0)
        if self.input_shape == "bnc":
            src = src.permute(0, 2, 1)
            tgt = tgt.permute(0, 2, 1)
        d_k = src_embedding.size(1)
        fused = d_k % 16 == 0 and not (torch.is_grad_enabled() and
                                       (src_embedding.requires_grad or tgt_embedding.requires_grad or tgt.requires_grad))
        if fused:
            src_corr = soft_correspondence(src_embedding, tgt_embedding, tgt)
        else:
            if src_embedding.is_cuda and all(z.dtype == torch.float32 for z in (src_embedding, tgt_embedding, tgt)) \
                    and tgt_embedding.size(2) <= 8192:
                # autograd live: the reference's op sequence (:27-31) on l3d_bmm_f32 / l3d_softmax_rows, forward and backward,
                # the transposed operands read through their strides (models/_rows.py)
                from ..models import _rows
                scores = _rows.softmax_rows(_rows.matmul(src_embedding.transpose(2, 1), tgt_embedding), 1.0 / math.sqrt(d_k))
                src_corr = _rows.matmul(tgt, scores.transpose(2, 1))
            else:
                scores = torch.matmul(src_embedding.transpose(2, 1).contiguous(), tgt_embedding) / math.sqrt(d_k)
                scores = torch.softmax(scores, dim=2)
                src_corr = torch.matmul(tgt, scores.transpose(2, 1).contiguous())
        if torch.is_grad_enabled() and (src_corr.requires_grad or src.requires_grad):
            R, t = _KabschFunction.apply(src, src_corr)
        else:
            R, t = kabsch(src, src_corr)
        return R, t.view(batch_size, 3)
