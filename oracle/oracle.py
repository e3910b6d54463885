"""ctypes/numpy front end of liboracle.so plus the numpy/torch-CPU restatements
of the parts of the path that are pure tensor algebra in the reference.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations are relative to
/root/reference/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (gcc, -ffp-contract=off).  Building the checker is
    not using it; __graft_entry__.build() calls this."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- a1
def knn(xyz_bn3, k, return_pd=False):
    """utils/model_common_utils.py:3-9 on xyz [B,N,3] -> int64 [B,N,k]."""
    x = _f(xyz_bn3)
    B, N, _ = x.shape
    idx = np.empty((B, N, k), np.int64)
    pd = np.empty((B, N, k), np.float32)
    lib().orc_knn(_p(x), B, N, k, _p(idx), _p(pd))
    return (idx, pd) if return_pd else idx


def knn_pd(xyz_bn3):
    """Full [B,N,N] matrix of the negated expanded distances knn() ranks."""
    x = _f(xyz_bn3)
    B, N, _ = x.shape
    pd = np.empty((B, N, N), np.float32)
    lib().orc_knn_pd(_p(x), B, N, _p(pd))
    return pd


def assert_knn_equal_modulo_ties(idx, ref_idx, xyz_bn3):
    """Bit-identical indices wherever the ranked values are distinct; under exact fp32
    ties (where the reference's topk order is unspecified) the
    two index lists must still carry the identical value sequence."""
    idx = np.asarray(idx).astype(np.int64)
    ref_idx = np.asarray(ref_idx).astype(np.int64)
    assert idx.shape == ref_idx.shape
    pd = knn_pd(xyz_bn3)
    va = np.take_along_axis(pd, idx, axis=2)
    vb = np.take_along_axis(pd, ref_idx, axis=2)
    assert np.array_equal(va, vb), "ranked value sequences differ"
    diff = idx != ref_idx
    if diff.any():
        k = idx.shape[2]
        # every differing slot must sit in a run of equal values (the k-th may tie with the (k+1)-th)
        kth = np.sort(pd, axis=2)[:, :, ::-1][:, :, :min(k + 1, pd.shape[2])]
        left = np.concatenate([np.zeros_like(diff[..., :1]), va[..., 1:] == va[..., :-1]], axis=2)
        right = np.concatenate([va[..., :-1] == va[..., 1:], (va[..., -1:] == kth[..., -1:])], axis=2)
        assert (~diff | left | right).all(), "indices differ outside exact ties"
    return int(diff.sum())


def get_graph_feature(x_b3n, k=20):
    """utils/model_common_utils.py:132-156: [B,C,N] -> [B,2C,N,k] = cat(neighbour, centre)."""
    x = _f(x_b3n)
    B, Cc, N = x.shape
    xt = np.ascontiguousarray(x.transpose(0, 2, 1))          # [B,N,C]
    idx = knn(xt, k)
    nb = np.stack([xt[b][idx[b]] for b in range(B)])         # [B,N,k,C]
    ctr = np.broadcast_to(xt[:, :, None, :], nb.shape)
    return np.ascontiguousarray(np.concatenate([nb, ctr], axis=3).transpose(0, 3, 1, 2))


# --------------------------------------------------------------------------- a3
def square_distance(src, dst):
    s, d = _f(src), _f(dst)
    B, N, _ = s.shape
    M = d.shape[1]
    out = np.empty((B, N, M), np.float32)
    lib().orc_square_distance(_p(s), _p(d), B, N, M, _p(out))
    return out


# --------------------------------------------------------------------------- a4
def query_ball_point(radius, nsample, xyz, new_xyz, get_cnt=False):
    x, q = _f(xyz), _f(new_xyz)
    B, N, _ = x.shape
    S = q.shape[1]
    idx = np.empty((B, S, nsample), np.int64)
    cnt = np.empty((B, S), np.int64)
    lib().orc_query_ball_point(C.c_float(radius), nsample, _p(x), _p(q), B, N, S, _p(idx), _p(cnt))
    return (idx, cnt) if get_cnt else idx


def query_ball_point_itself(radius, nsample, xyz, new_xyz, itself_indices):
    """utils/ppfnet_util.py:96-131 with itself_indices: the centre's own index is taken out of its neighbourhood (:116-119, set to
    N before the radius test), the first nsample hits in index order are kept (sort, :122) and short rows are padded with the
    centre's index (:123-128).  numpy on top of square_distance (the expanded fp32 form the reference's square_distance computes)."""
    x, q = _f(xyz), _f(new_xyz)
    it = np.asarray(itself_indices, dtype=np.int64)
    B, N, _ = x.shape
    S = q.shape[1]
    d2 = square_distance(q, x)
    idx = np.empty((B, S, nsample), np.int64)
    r2 = np.float32(radius ** 2)                              # `radius ** 2` is a Python float; the comparison promotes it to fp32
    for b in range(B):
        for s in range(S):
            hits = np.nonzero(~(d2[b, s] > r2))[0]
            hits = hits[hits != it[b, s]][:nsample]
            idx[b, s, :len(hits)] = hits
            idx[b, s, len(hits):] = it[b, s]
    return idx


# --------------------------------------------------------------------------- a5
def index_points(points, idx):
    """utils/model_common_utils.py:40-56."""
    points = np.asarray(points)
    idx = np.asarray(idx)
    B = points.shape[0]
    return np.stack([points[b][idx[b]] for b in range(B)])


# --------------------------------------------------------------------------- a6
def farthest_point_sample(xyz, npoint):
    """start_with_first_point=True variant (the random start is not testable)."""
    x = _f(xyz)
    B, N, _ = x.shape
    out = np.empty((B, npoint), np.int64)
    lib().orc_farthest_point_sample(_p(x), B, N, npoint, _p(out))
    return out


# --------------------------------------------------------------------------- a7
def knn_point(k, pos1, pos2):
    p1, p2 = _f(pos1), _f(pos2)
    B, N, _ = p1.shape
    M = p2.shape[1]
    val = np.empty((B, M, k), np.float32)
    idx = np.empty((B, M, k), np.int64)
    lib().orc_knn_point(k, _p(p1), _p(p2), B, N, M, _p(val), _p(idx))
    return val, idx


def knn_point_expanded(nsample, xyz, new_xyz):
    """utils/pointconv_util.py:107-118: the nsample smallest entries of square_distance(new_xyz, xyz) per query.
    torch.topk(sorted=False) leaves the row order open; rows are returned nearest first, lowest index first
    under exact ties (stable argsort), which is also what the HIP kernel emits."""
    d = square_distance(new_xyz, xyz)                       # the reference's expanded fp32 rounding sequence
    return np.argsort(d, axis=-1, kind="stable")[..., :nsample].astype(np.int64)


# --------------------------------------------------------------------------- a9
def chamfer_forward(xyz1, xyz2):
    a, b = _f(xyz1), _f(xyz2)
    B, N, _ = a.shape
    M = b.shape[1]
    d1 = np.empty((B, N), np.float32); d2 = np.empty((B, M), np.float32)
    i1 = np.empty((B, N), np.int32); i2 = np.empty((B, M), np.int32)
    lib().orc_chamfer_forward(_p(a), _p(b), B, N, M, _p(d1), _p(d2), _p(i1), _p(i2))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, gd1, gd2, idx1, idx2):
    a, b = _f(xyz1), _f(xyz2)
    B, N, _ = a.shape
    M = b.shape[1]
    g1 = np.empty_like(a); g2 = np.empty_like(b)
    gd1, gd2, idx1, idx2 = _f(gd1), _f(gd2), _i32(idx1), _i32(idx2)
    lib().orc_chamfer_backward(_p(a), _p(b), B, N, M, _p(gd1), _p(gd2), _p(idx1), _p(idx2), _p(g1), _p(g2))
    return g1, g2


def chamfer_loss(template, source):
    """losses/chamfer_distance.py:34-43: (mean sqrt d1 + mean sqrt d2) / 2 over the whole batch."""
    d1, d2, _, _ = chamfer_forward(template, source)
    m1 = np.sqrt(d1).astype(np.float32).mean(dtype=np.float64)
    m2 = np.sqrt(d2).astype(np.float32).mean(dtype=np.float64)
    return np.float32((m1 + m2) / 2.0)


# --------------------------------------------------------------------------- K7-K16
def ball_query(radius, nsample, xyz, new_xyz):
    x, q = _f(xyz), _f(new_xyz)
    B, N, _ = x.shape
    S = q.shape[1]
    idx = np.zeros((B, S, nsample), np.int32)
    lib().orc_ball_query(B, N, S, C.c_float(radius), nsample, _p(q), _p(x), _p(idx))
    return idx


def group_points(points_bcn, idx_bsk):
    p, ix = _f(points_bcn), _i32(idx_bsk)
    B, Cc, N = p.shape
    _, S, K = ix.shape
    out = np.empty((B, Cc, S, K), np.float32)
    lib().orc_group_points(B, Cc, N, S, K, _p(p), _p(ix), _p(out))
    return out


def group_points_grad(grad_out, idx_bsk, N):
    g, ix = _f(grad_out), _i32(idx_bsk)
    B, Cc, S, K = g.shape
    out = np.empty((B, Cc, N), np.float32)
    lib().orc_group_points_grad(B, Cc, N, S, K, _p(g), _p(ix), _p(out))
    return out


def gather_points(points_bcn, idx_bs):
    p, ix = _f(points_bcn), _i32(idx_bs)
    B, Cc, N = p.shape
    S = ix.shape[1]
    out = np.empty((B, Cc, S), np.float32)
    lib().orc_gather_points(B, Cc, N, S, _p(p), _p(ix), _p(out))
    return out


def gather_points_grad(grad_out, idx_bs, N):
    g, ix = _f(grad_out), _i32(idx_bs)
    B, Cc, S = g.shape
    out = np.empty((B, Cc, N), np.float32)
    lib().orc_gather_points_grad(B, Cc, N, S, _p(g), _p(ix), _p(out))
    return out


def furthest_point_sampling(xyz, npoint):
    x = _f(xyz)
    B, N, _ = x.shape
    temp = np.full((B, N), 1e10, np.float32)
    out = np.empty((B, npoint), np.int32)
    lib().orc_furthest_point_sampling(B, N, npoint, _p(x), _p(temp), _p(out))
    return out


def knn_pair(k, unknown, known):
    """pointnet2_utils.knn: returns (sqrt(dist2), idx) like pointnet2_utils.py:94-95."""
    u, kn = _f(unknown), _f(known)
    B, N, _ = u.shape
    M = kn.shape[1]
    d2 = np.empty((B, N, k), np.float32)
    idx = np.empty((B, N, k), np.int32)
    lib().orc_knn_pair(B, N, M, k, _p(u), _p(kn), _p(d2), _p(idx))
    return np.sqrt(d2), idx


def three_nn(unknown, known):
    return knn_pair(3, unknown, known)


def three_interpolate(points_bcm, idx_bn3, weight_bn3):
    p, ix, w = _f(points_bcm), _i32(idx_bn3), _f(weight_bn3)
    B, Cc, M = p.shape
    N = ix.shape[1]
    out = np.empty((B, Cc, N), np.float32)
    lib().orc_three_interpolate(B, Cc, M, N, _p(p), _p(ix), _p(w), _p(out))
    return out


def three_interpolate_grad(grad_out, idx_bn3, weight_bn3, M):
    g, ix, w = _f(grad_out), _i32(idx_bn3), _f(weight_bn3)
    B, Cc, N = g.shape
    out = np.empty((B, Cc, M), np.float32)
    lib().orc_three_interpolate_grad(B, Cc, N, M, _p(g), _p(ix), _p(w), _p(out))
    return out


# --------------------------------------------------------------------------- a10
def emd_forward(xyz1, xyz2):
    a, b = _f(xyz1), _f(xyz2)
    B, n, _ = a.shape
    m = b.shape[1]
    match = np.empty((B, n, m), np.float32)      # allocated [B,n,m], indexed [l*n+k] (emd.cu:18, emd.cuh:158)
    cost = np.empty((B,), np.float32)
    lib().orc_emd_approxmatch(B, n, m, _p(a), _p(b), _p(match))
    lib().orc_emd_matchcost(B, n, m, _p(a), _p(b), _p(match), _p(cost))
    return cost, match


def emd_backward(xyz1, xyz2, match):
    a, b, mt = _f(xyz1), _f(xyz2), _f(match)
    B, n, _ = a.shape
    m = b.shape[1]
    g1 = np.empty_like(a); g2 = np.empty_like(b)
    lib().orc_emd_matchcostgrad(B, n, m, _p(a), _p(b), _p(mt), _p(g1), _p(g2))
    return g1, g2


# --------------------------------------------------------------------------- a11
def svd_head(src_emb, tgt_emb, src_bn3, tgt_bn3):
    """utils/svd.py:13-59 in float32 numpy (np.linalg.svd = LAPACK gesdd, as torch.svd on CPU).

    src_emb/tgt_emb [B,C,N]; src/tgt [B,N,3] ("bnc").  Returns R [B,3,3], t [B,3]."""
    se, te = _f(src_emb), _f(tgt_emb)
    src = _f(src_bn3).transpose(0, 2, 1)       # [B,3,N]
    tgt = _f(tgt_bn3).transpose(0, 2, 1)
    d_k = se.shape[1]
    scores = np.matmul(se.transpose(0, 2, 1), te) / np.float32(np.sqrt(d_k))
    scores = scores - scores.max(axis=2, keepdims=True)
    e = np.exp(scores)
    scores = (e / e.sum(axis=2, keepdims=True)).astype(np.float32)
    src_corr = np.matmul(tgt, scores.transpose(0, 2, 1))
    return kabsch(src, src_corr)


def kabsch(src_b3n, corr_b3n):
    """utils/svd.py:29-58: centre, H = src_c corr_c^T, R = V U^T (det-fixed), t."""
    src, corr = _f(src_b3n), _f(corr_b3n)
    sm = src.mean(axis=2, keepdims=True, dtype=np.float32)
    cm = corr.mean(axis=2, keepdims=True, dtype=np.float32)
    H = np.matmul(src - sm, (corr - cm).transpose(0, 2, 1))
    R = rotation_from_H(H)
    t = np.matmul(-R, sm) + cm
    return R.astype(np.float32), t[:, :, 0].astype(np.float32)


def rotation_from_H(H):
    """utils/svd.py:38-49 per item: u,s,v = svd(H); r = v u^T; if det(r) < 0: v = v diag(1,1,-1)."""
    H = np.asarray(H)
    R = np.empty_like(H)
    refl = np.diag([1.0, 1.0, -1.0]).astype(H.dtype)
    for i in range(H.shape[0]):
        u, s, vt = np.linalg.svd(H[i])
        v = vt.T
        r = v @ u.T
        if np.linalg.det(r) < 0:
            r = (v @ refl) @ u.T
        R[i] = r
    return R


# --------------------------------------------------------------------------- a8
def dgcnn_forward_torch(x_bn3, weights, k=20, eps=1e-5):
    """models/dgcnn.py:25-49 in eval mode, restated with plain torch CPU ops.

    `weights` is a dict with conv{1..5}.weight [Co,Ci,1,1] and bn{1..5}.{weight,bias,
    running_mean,running_var}.  The graph comes from this oracle's own knn (C) so the
    whole forward is independent of learning3d_amd."""
    import torch
    import torch.nn.functional as F
    x = torch.as_tensor(np.asarray(x_bn3, dtype=np.float32))
    B, N, _ = x.shape
    idx = torch.from_numpy(knn(x.numpy(), k))                       # [B,N,k]
    nb = torch.gather(x.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, k, 3))
    feat = torch.cat([nb, x.unsqueeze(2).expand(B, N, k, 3)], dim=3).permute(0, 3, 1, 2)  # [B,6,N,k]

    def block(h, i):
        w = torch.as_tensor(weights[f"conv{i}.weight"])
        h = F.conv2d(h, w)
        h = F.batch_norm(h, torch.as_tensor(weights[f"bn{i}.running_mean"]),
                         torch.as_tensor(weights[f"bn{i}.running_var"]),
                         torch.as_tensor(weights[f"bn{i}.weight"]),
                         torch.as_tensor(weights[f"bn{i}.bias"]), False, 0.0, eps)
        return F.relu(h)

    outs = []
    h = feat
    for i in (1, 2, 3, 4):
        h = block(h, i)
        outs.append(h.max(dim=-1, keepdim=True)[0])
    h = block(torch.cat(outs, dim=1), 5)
    return h.view(B, -1, N)


def dgcnn_forward_refops(x_bn3, weights, k=20, eps=1e-5):
    """The reference's OWN op sequence for models/dgcnn.py:25-49 on torch CPU, all threads -- the CPU baseline bench.py times
    (kind "reference-op-sequence"); dgcnn_forward_torch above is the checker (C kNN with the documented tie order).
      knn               utils/model_common_utils.py:3-9    -2 * matmul(x^T, x), sum(x**2), -xx - inner - xx^T, topk
      get_graph_feature utils/model_common_utils.py:132-156  idx + idx_base, flat row gather, repeat, cat, permute
      conv/bn/relu/max  models/dgcnn.py:34-46, cat :46, conv5 :48
    Module objects are replaced by their functional forms (F.conv2d / F.batch_norm in eval mode: the same ATen kernels)."""
    import torch
    import torch.nn.functional as F
    x = torch.as_tensor(np.asarray(x_bn3, dtype=np.float32)).permute(0, 2, 1)      # "bnc" -> [B,3,N], dgcnn.py:26-27
    B, C, N = x.shape
    inner = -2 * torch.matmul(x.transpose(2, 1).contiguous(), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pd = -xx - inner - xx.transpose(2, 1).contiguous()
    idx = pd.topk(k=k, dim=-1)[1]
    idx = (idx + torch.arange(0, B).view(-1, 1, 1) * N).view(-1)
    xt = x.transpose(2, 1).contiguous()
    feature = xt.view(B * N, -1)[idx, :].view(B, N, k, C)
    xr = xt.view(B, N, 1, C).repeat(1, 1, k, 1)
    h = torch.cat((feature, xr), dim=3).permute(0, 3, 1, 2)

    def block(h, i):
        h = F.conv2d(h, torch.as_tensor(weights[f"conv{i}.weight"]))
        h = F.batch_norm(h, torch.as_tensor(weights[f"bn{i}.running_mean"]), torch.as_tensor(weights[f"bn{i}.running_var"]),
                         torch.as_tensor(weights[f"bn{i}.weight"]), torch.as_tensor(weights[f"bn{i}.bias"]), False, 0.0, eps)
        return F.relu(h)

    outs = []
    for i in (1, 2, 3, 4):
        h = block(h, i)
        outs.append(h.max(dim=-1, keepdim=True)[0])
    return block(torch.cat(outs, dim=1), 5).view(B, -1, N)


def chamfer_loss_refops(template, source):
    """losses/chamfer_distance.py:5-31, the reference's torch fallback (what its `except:` branch runs on a CPU host, where the
    `cuda.chamfer_distance` extension cannot build): the [m,n,n,3] broadcast difference, abs, pow 2, sum, min over both axes,
    sqrt, means.  bench.py's CPU baseline; chamfer_loss above (C nnsearch) is the checker."""
    import torch
    a = torch.as_tensor(np.asarray(template, dtype=np.float32))
    b = torch.as_tensor(np.asarray(source, dtype=np.float32))
    M = (a.unsqueeze(2) - b.unsqueeze(1)).abs().pow(2).sum(3)
    return (torch.mean(torch.sqrt(M.min(1)[0])) + torch.mean(torch.sqrt(M.min(2)[0]))) / 2.0


def knn_feature(x_bcn, k):
    """utils/model_common_utils.py:3-9 for a feature map x [B,C,N] (any C): the reference's own op sequence in
    torch CPU fp32 (matmul -> MKL sgemm, topk) -- what models/prnet.py:76-97 calls per layer."""
    import torch
    x = torch.as_tensor(np.asarray(x_bcn, dtype=np.float32))
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pd = -xx - inner - xx.transpose(2, 1)
    return pd.topk(k=k, dim=-1)[1]


def prnet_dgcnn_forward_torch(x_b3n, weights, k=20, eps=1e-5, slope=0.2):
    """models/prnet.py:62-97 (class DGCNN) in eval mode, restated with plain torch CPU ops: every layer rebuilds
    the k-NN graph in the feature space of the previous layer's output (get_graph_feature on x, x1, x2, x3)."""
    import torch
    import torch.nn.functional as F
    x = torch.as_tensor(np.asarray(x_b3n, dtype=np.float32))
    B, _, N = x.shape

    def graph_feature(h):                                              # model_common_utils.py:132-156
        C = h.shape[1]
        idx = knn_feature(h.numpy(), k)                                # [B,N,k]
        ht = h.transpose(2, 1)                                         # [B,N,C]
        nb = torch.gather(ht.unsqueeze(1).expand(B, N, N, C), 2, idx.unsqueeze(-1).expand(B, N, k, C))
        return torch.cat([nb, ht.unsqueeze(2).expand(B, N, k, C)], dim=3).permute(0, 3, 1, 2)

    def block(h, i):
        h = F.conv2d(h, torch.as_tensor(weights[f"conv{i}.weight"]))
        h = F.batch_norm(h, torch.as_tensor(weights[f"bn{i}.running_mean"]), torch.as_tensor(weights[f"bn{i}.running_var"]),
                         torch.as_tensor(weights[f"bn{i}.weight"]), torch.as_tensor(weights[f"bn{i}.bias"]), False, 0.0, eps)
        return F.leaky_relu(h, negative_slope=slope)

    outs, h = [], x
    for i in (1, 2, 3, 4):
        h = block(graph_feature(h), i).max(dim=-1, keepdim=True)[0]    # [B,Co,N,1]
        outs.append(h)
        h = h.view(B, -1, N)
    return block(torch.cat(outs, dim=1), 5).view(B, -1, N)


# ----------------------------------------------------------------- a8: PCN, PointNet classifier (config 1)
def _t(a):
    import torch
    return torch.as_tensor(np.asarray(a))


def pcn_forward_torch(x_bn3, w, num_coarse, grid_size):
    """models/pcn.py:104-153 (encode :104-119, decode :121-126, fine_decoder :84-102) in eval mode, restated with
    plain torch CPU functionals.  `w`: conv{1..7}.{weight,bias}, linear{1..3}.{weight,bias} (the reference's
    state_dict keys)."""
    import torch
    import torch.nn.functional as F
    x = _t(x_bn3).float().permute(0, 2, 1)                                        # [B,3,N]
    B, _, N = x.shape
    c = lambda h, i: F.conv1d(h, _t(w[f"conv{i}.weight"]), _t(w[f"conv{i}.bias"]))
    l = lambda h, i: F.linear(h, _t(w[f"linear{i}.weight"]), _t(w[f"linear{i}.bias"]))
    h = c(F.relu(c(x, 1)), 2)                                                      # encoder_1
    g = h.max(dim=2)[0]
    h = torch.cat([h, g.unsqueeze(2).repeat(1, 1, N)], dim=1)
    h = c(F.relu(c(h, 3)), 4)                                                      # encoder_2
    gv = h.max(dim=2)[0]                                                           # [B,1024]
    coarse = l(F.relu(l(F.relu(l(gv, 1)), 2)), 3).view(B, num_coarse, 3)
    num_fine = grid_size ** 2 * num_coarse
    lin = torch.linspace(-0.05, 0.05, steps=grid_size)
    grid = torch.reshape(torch.stack(torch.meshgrid(lin, lin, indexing="ij"), dim=2), (-1, 2)).unsqueeze(0)
    grid_feature = grid.repeat([B, num_coarse, 1])
    point_feature = coarse.unsqueeze(2).repeat([1, 1, grid_size ** 2, 1]).reshape(-1, num_fine, 3)
    global_feature = gv.unsqueeze(1).repeat([1, num_fine, 1])
    feature = torch.cat([grid_feature, point_feature, global_feature], dim=2).permute(0, 2, 1)
    out = c(F.relu(c(F.relu(c(feature, 5)), 6)), 7)
    return coarse.numpy(), (out.permute(0, 2, 1) + point_feature).numpy()


# ----------------------------------------------------------------- a12 as a model: FlowNet3D (config 5)
def _cbr(h, w, pre, eps=1e-5):
    """Conv(1x1, optional bias) -> BatchNorm (eval) -> ReLU with the state_dict keys pre.{weight | bias | running_*};
    `pre` = (conv key prefix, bn key prefix).  h [B,C,N] or [B,C,S,K]."""
    import torch.nn.functional as F
    ck, bk = pre
    wt = _t(w[ck + ".weight"])
    b = _t(w[ck + ".bias"]) if ck + ".bias" in w else None
    h = F.conv2d(h, wt, b) if h.dim() == 4 else F.conv1d(h, wt, b)
    h = F.batch_norm(h, _t(w[bk + ".running_mean"]), _t(w[bk + ".running_var"]), _t(w[bk + ".weight"]), _t(w[bk + ".bias"]),
                     False, 0.0, eps)
    return F.relu(h)


def set_abstraction_forward_torch(xyz_b3n, feat_bcn, w, prefix, npoint, radius, nsample, n_layers=3):
    """PointNetSetAbstraction.forward, models/flownet3d.py:93-123: furthest_point_sample (K12) -> gather_operation (K10)
    -> QueryAndGroup = ball_query (K7) + grouping_operation (K8) - centre, concat [xyz | features]
    (utils/lib/pointnet2_utils.py:259-292) -> [Conv2d + BN + ReLU] * n -> max over the nsample axis.
    `w`: state_dict with keys prefix.mlp_convs.i.weight / prefix.mlp_bns.i.*; prefix may be "" for a bare layer."""
    import torch
    xyz = np.ascontiguousarray(np.asarray(xyz_b3n, np.float32))
    feat = np.ascontiguousarray(np.asarray(feat_bcn, np.float32))
    xyz_t = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    fps = furthest_point_sampling(xyz_t, npoint)
    new_xyz = gather_points(xyz, fps)                                               # [B,3,S]
    idx = ball_query(radius, nsample, xyz_t, np.ascontiguousarray(new_xyz.transpose(0, 2, 1)))
    g_xyz = group_points(xyz, idx) - new_xyz[:, :, :, None]
    h = torch.from_numpy(np.concatenate([g_xyz, group_points(feat, idx)], axis=1))
    pre = prefix + "." if prefix else ""
    with torch.no_grad():
        for i in range(n_layers):
            h = _cbr(h, w, (f"{pre}mlp_convs.{i}", f"{pre}mlp_bns.{i}"))
        return new_xyz, h.max(-1)[0].numpy()


def flownet3d_forward_torch(pc1, pc2, feature1, feature2, w, return_intermediates=False):
    """FlowNet3D.forward, models/flownet3d.py:305-328 (layer hyper-parameters :293-303), eval mode, composed from the K7-K16
    restatements of oracle.c and torch-CPU conv / BatchNorm functionals: PointNetSetAbstraction :93-123 (above),
    FlowEmbedding :142-180 (knn branch), PointNetSetUpConv :208-242 (knn branch), PointNetFeaturePropogation :256-286.
    `w`: the reference FlowNet3D's state_dict (numpy values)."""
    import torch
    pc1, pc2 = np.asarray(pc1, np.float32), np.asarray(pc2, np.float32)
    f1, f2 = np.asarray(feature1, np.float32), np.asarray(feature2, np.float32)
    tr = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1))
    sa = lambda x, f, name, S, r, K: set_abstraction_forward_torch(x, f, w, name, S, r, K)

    def flow_embedding(pos1, pos2, feat1, feat2, nsample=64):                      # :142-180, self.knn = True
        _, idx = knn_pair(nsample, tr(pos1), tr(pos2))
        pos_diff = group_points(pos2, idx) - pos1[:, :, :, None]
        feat_diff = np.concatenate([group_points(feat2, idx), np.repeat(feat1[:, :, :, None], nsample, axis=3)], axis=1)
        h = torch.from_numpy(np.concatenate([pos_diff, feat_diff], axis=1))
        for i in range(3):
            h = _cbr(h, w, (f"fe_layer.mlp_convs.{i}", f"fe_layer.mlp_bns.{i}"))
        return h.max(-1)[0].numpy()

    def set_upconv(name, pos1, pos2, feat1, feat2, n1, n2, nsample=8):              # :208-242
        _, idx = knn_pair(nsample, tr(pos1), tr(pos2))
        pos_diff = group_points(pos2, idx) - pos1[:, :, :, None]
        h = torch.from_numpy(np.concatenate([group_points(feat2, idx), pos_diff], axis=1))
        for i in range(n1):
            h = _cbr(h, w, (f"{name}.mlp1_convs.{i}.0", f"{name}.mlp1_convs.{i}.1"))
        h = h.max(-1)[0]
        if feat1 is not None:
            h = torch.cat([h, torch.from_numpy(feat1)], dim=1)
        for i in range(n2):
            h = _cbr(h, w, (f"{name}.mlp2_convs.{i}.0", f"{name}.mlp2_convs.{i}.1"))
        return h.numpy()

    def feature_propagation(pos1, pos2, feat1, feat2):                              # :256-286
        dists, idx = three_nn(tr(pos1), tr(pos2))
        dists = np.where(dists < 1e-10, np.float32(1e-10), dists).astype(np.float32)
        weight = (np.float32(1.0) / dists).astype(np.float32)
        weight = (weight / weight.sum(-1, keepdims=True)).astype(np.float32)
        g = torch.from_numpy(group_points(feat2, idx)) * torch.from_numpy(weight)[:, None]     # [B,C,N,3]
        h = torch.cat([torch.sum(g, dim=-1), torch.from_numpy(feat1)], dim=1)
        for i in range(2):
            h = _cbr(h, w, (f"fp.mlp_convs.{i}", f"fp.mlp_bns.{i}"))
        return h

    with torch.no_grad():
        l1_pc1, l1_f1 = sa(pc1, f1, "sa1", 1024, 0.5, 16)
        l2_pc1, l2_f1 = sa(l1_pc1, l1_f1, "sa2", 256, 1.0, 16)
        l1_pc2, l1_f2 = sa(pc2, f2, "sa1", 1024, 0.5, 16)
        l2_pc2, l2_f2 = sa(l1_pc2, l1_f2, "sa2", 256, 1.0, 16)
        l2_f1_new = flow_embedding(l2_pc1, l2_pc2, l2_f1, l2_f2)
        l3_pc1, l3_f1 = sa(l2_pc1, l2_f1_new, "sa3", 64, 2.0, 8)
        l4_pc1, l4_f1 = sa(l3_pc1, l3_f1, "sa4", 16, 4.0, 8)
        l3_fnew1 = set_upconv("su1", l3_pc1, l4_pc1, l3_f1, l4_f1, 0, 2)
        l2_fnew1 = set_upconv("su2", l2_pc1, l3_pc1, np.concatenate([l2_f1, l2_f1_new], axis=1), l3_fnew1, 3, 1)
        l1_fnew1 = set_upconv("su3", l1_pc1, l2_pc1, l1_f1, l2_fnew1, 3, 1)
        l0 = feature_propagation(pc1, l1_pc1, f1, l1_fnew1)
        x = _cbr(l0, w, ("conv1", "bn1"))
        sf = torch.nn.functional.conv1d(x, _t(w["conv2.weight"]), _t(w["conv2.bias"])).numpy()
    if return_intermediates:
        return sf, dict(l1_pc1=l1_pc1, l1_feature1=l1_f1, l2_feature1=l2_f1, l2_feature1_new=l2_f1_new)
    return sf


def pointnet_classifier_forward_torch(x_bn3, w, eps=1e-5):
    """models/pointnet.py:45-73 (use_bn=True) -> Pooling('max') -> models/classifier.py:22-29, eval mode (dropout is
    the identity), restated with torch CPU functionals.  `w` has the checkpoint's keys (feature_model.*, linear*, bn*)."""
    import torch.nn.functional as F
    h = _t(x_bn3).float().permute(0, 2, 1)
    bn = lambda h, p: F.batch_norm(h, _t(w[p + ".running_mean"]), _t(w[p + ".running_var"]), _t(w[p + ".weight"]),
                                   _t(w[p + ".bias"]), False, 0.0, eps)
    for i in range(1, 6):
        h = F.relu(bn(F.conv1d(h, _t(w[f"feature_model.conv{i}.weight"]), _t(w[f"feature_model.conv{i}.bias"])),
                      f"feature_model.bn{i}"))
    h = h.max(dim=2)[0]
    h = F.relu(bn(F.linear(h, _t(w["linear1.weight"]), _t(w["linear1.bias"])), "bn1"))
    h = F.relu(bn(F.linear(h, _t(w["linear2.weight"]), _t(w["linear2.bias"])), "bn2"))
    return F.linear(h, _t(w["linear3.weight"]), _t(w["linear3.bias"])).numpy()


def gaussian_density(xyz, bandwidth):
    """utils/pointconv_util.py:194-203: mean_j exp(-square_distance / (2 bw^2)) / (2.5 bw), fp32 like the reference
    (the two divisors are Python doubles that torch narrows to fp32 scalars)."""
    d = square_distance(xyz, xyz)
    g = np.exp(-d / np.float32(2.0 * bandwidth * bandwidth)) / np.float32(2.5 * bandwidth)
    return g.astype(np.float32).mean(axis=-1, dtype=np.float32)


# ----------------------------------------------------------------- config 3: DCP-v2 (DGCNN + Transformer + SVD head)
def dcp_forward_torch(template, source, w, n_heads=4, k=20, dtype="float32"):
    """models/dcp.py:30-56 with pointer_='transformer', head='svd', cycle=False, eval mode, restated as torch CPU
    functionals: emb_nn = DGCNN (models/dgcnn.py:25-49), pointer = Transformer(emb, 1 block, ff 1024, 4 heads)
    (utils/transformer.py:17-25, :94-216: unbiased-std LayerNorm with eps on the std, pre-norm residual sublayers,
    encoder(src) -> decoder(tgt, memory), called twice with the roles swapped :238-245), head = SVDHead
    (utils/svd.py:13-59).  `w` has the DCP state_dict keys (emb_nn.*, pointer.model.*, head.reflect).
    dtype="float64" evaluates the same graph in double (kNN graph still from the fp32 oracle) -- used to measure how
    far fp32 rounding alone moves R, t on a given input."""
    import math
    import torch
    import torch.nn.functional as F
    dt = getattr(torch, dtype)
    W = lambda name: torch.as_tensor(np.asarray(w[name])).to(dt)

    def dgcnn(x_bn3):
        B, N, _ = x_bn3.shape
        idx = torch.from_numpy(knn(np.asarray(x_bn3, dtype=np.float32), k))
        x = torch.as_tensor(np.asarray(x_bn3)).to(dt)
        nb = torch.gather(x.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, k, 3))
        h = torch.cat([nb, x.unsqueeze(2).expand(B, N, k, 3)], dim=3).permute(0, 3, 1, 2)
        outs = []
        for i in (1, 2, 3, 4, 5):
            if i == 5:
                h = torch.cat(outs, dim=1)
            h = F.conv2d(h, W(f"emb_nn.conv{i}.weight"))
            h = F.relu(F.batch_norm(h, W(f"emb_nn.bn{i}.running_mean"), W(f"emb_nn.bn{i}.running_var"),
                                    W(f"emb_nn.bn{i}.weight"), W(f"emb_nn.bn{i}.bias"), False, 0.0, 1e-5))
            if i < 5:
                outs.append(h.max(dim=-1, keepdim=True)[0])          # the un-pooled h feeds the next conv (:36-46)
        return h.view(B, -1, N)

    def layer_norm(x, p):
        mean, std = x.mean(-1, keepdim=True), x.std(-1, keepdim=True)
        return W(p + ".a_2") * (x - mean) / (std + 1e-6) + W(p + ".b_2")

    def mha(q, k_, v, p):
        nb = q.size(0)
        lin = lambda x, i: F.linear(x, W(f"{p}.linears.{i}.weight"), W(f"{p}.linears.{i}.bias"))
        d_model = q.size(-1)
        d_k = d_model // n_heads
        q, k_, v = [lin(x, i).view(nb, -1, n_heads, d_k).transpose(1, 2) for i, x in enumerate((q, k_, v))]
        scores = torch.matmul(q, k_.transpose(-2, -1)) / math.sqrt(d_k)
        x = torch.matmul(F.softmax(scores, dim=-1), v)
        return lin(x.transpose(1, 2).contiguous().view(nb, -1, d_model), 3)

    def ff(x, p):
        return F.linear(F.relu(F.linear(x, W(p + ".w_1.weight"), W(p + ".w_1.bias"))), W(p + ".w_2.weight"), W(p + ".w_2.bias"))

    def pointer_pass(src, tgt):                      # self.model(src, tgt, None, None)
        e = "pointer.model.encoder"
        x = src
        x = x + mha(*([layer_norm(x, e + ".layers.0.sublayer.0.norm")] * 3), e + ".layers.0.self_attn")
        x = x + ff(layer_norm(x, e + ".layers.0.sublayer.1.norm"), e + ".layers.0.feed_forward")
        mem = layer_norm(x, e + ".norm")
        d = "pointer.model.decoder"
        x = tgt
        x = x + mha(*([layer_norm(x, d + ".layers.0.sublayer.0.norm")] * 3), d + ".layers.0.self_attn")
        x = x + mha(layer_norm(x, d + ".layers.0.sublayer.1.norm"), mem, mem, d + ".layers.0.src_attn")
        x = x + ff(layer_norm(x, d + ".layers.0.sublayer.2.norm"), d + ".layers.0.feed_forward")
        return layer_norm(x, d + ".norm")

    sf, tf = dgcnn(source), dgcnn(template)                                    # [B,C,N]
    s_t, t_t = sf.transpose(2, 1).contiguous(), tf.transpose(2, 1).contiguous()
    tgt_emb = pointer_pass(s_t, t_t).transpose(2, 1)                           # Transformer.forward(src=sf, tgt=tf)
    src_emb = pointer_pass(t_t, s_t).transpose(2, 1)
    sf, tf = sf + src_emb, tf + tgt_emb
    src = torch.as_tensor(np.asarray(source)).to(dt).permute(0, 2, 1)
    tgt = torch.as_tensor(np.asarray(template)).to(dt).permute(0, 2, 1)
    B = src.shape[0]
    d_k = sf.size(1)
    scores = torch.softmax(torch.matmul(sf.transpose(2, 1).contiguous(), tf) / math.sqrt(d_k), dim=2)
    corr = torch.matmul(tgt, scores.transpose(2, 1).contiguous())
    sc, cc = src - src.mean(dim=2, keepdim=True), corr - corr.mean(dim=2, keepdim=True)
    H = torch.matmul(sc, cc.transpose(2, 1).contiguous())
    refl = torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=dt))
    Rs = []
    for i in range(B):
        u, s_, v = torch.svd(H[i])
        r = torch.matmul(v, u.transpose(1, 0))
        if torch.det(r) < 0:
            r = torch.matmul(torch.matmul(v, refl), u.transpose(1, 0))
        Rs.append(r)
    R = torch.stack(Rs, dim=0)
    t = (torch.matmul(-R, src.mean(dim=2, keepdim=True)) + corr.mean(dim=2, keepdim=True)).view(B, 3)
    return {"est_R": R.numpy(), "est_t": t.numpy(), "r": (tf - sf).numpy(), "H": H.numpy()}


# ----------------------------------------------------------------- 8(f) rank 4: DCPTransform
def dcp_transform(template, anglex, angley, anglez, translation):
    """ops/transform_functions.py:304-310 restated without scipy: Rotation.from_euler('zyx', [az, ay, ax]) is the
    extrinsic z, y, x sequence, i.e. R = Rx(ax) Ry(ay) Rz(az); source = template R^T + t; igt = [R^T | t ; 0 0 0 1]
    (the reference stores Rotation.apply(np.eye(3)), which is R^T).  fp64 like scipy, rounded to fp32 at the end."""
    t = np.asarray(template, np.float64)
    B = t.shape[0]
    src = np.empty_like(t)
    igt = np.zeros((B, 4, 4))
    for b in range(B):
        cx, sx = np.cos(anglex[b]), np.sin(anglex[b])
        cy, sy = np.cos(angley[b]), np.sin(angley[b])
        cz, sz = np.cos(anglez[b]), np.sin(anglez[b])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = Rx @ Ry @ Rz
        src[b] = t[b] @ R.T + np.asarray(translation[b])[None]
        igt[b, :3, :3] = R.T
        igt[b, :3, 3] = translation[b]
        igt[b, 3, 3] = 1.0
    return src.astype(np.float32), igt.astype(np.float32)


def twist_transform(template, twist):
    """ops/transform_functions.py:133-141 + ops/se3.py:51-74 + ops/sinc.py (PNLKTransform.apply_transform) in fp64:
    template [B,N,3], twist [B,6] -> (source, igt = se3.exp(x), gt = se3.exp(-x)), rounded to fp32 once."""
    t = np.asarray(template, np.float64)
    x = np.asarray(twist, np.float64)

    def exp(xx):
        out = np.zeros((xx.shape[0], 4, 4))
        for b, (w, v) in enumerate(zip(xx[:, :3], xx[:, 3:])):
            th2 = float(w @ w)
            th = np.sqrt(th2)
            if th < 0.01:
                s1 = 1 - th2 / 6 * (1 - th2 / 20 * (1 - th2 / 42))
                s2 = 0.5 * (1 - th2 / 12 * (1 - th2 / 30 * (1 - th2 / 56)))
                s3 = 1 / 6 * (1 - th2 / 20 * (1 - th2 / 42 * (1 - th2 / 72)))
            else:
                s1, s2, s3 = np.sin(th) / th, (1 - np.cos(th)) / th2, (th - np.sin(th)) / th ** 3
            W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            S = W @ W
            out[b, :3, :3] = np.eye(3) + s1 * W + s2 * S
            out[b, :3, 3] = (np.eye(3) + s2 * W + s3 * S) @ v
            out[b, 3, 3] = 1
        return out
    g, gt = exp(x), exp(-x)
    src = np.einsum("bij,bnj->bni", g[:, :3, :3], t) + g[:, None, :3, 3]
    return src.astype(np.float32), g.astype(np.float32), gt.astype(np.float32)


def quat_transform(template, pose7):
    """PCRNetTransform.__call__ (ops/transform_functions.py:218-269, ops/quaternion.py:35-53) in fp32 torch ops."""
    import torch
    t = torch.from_numpy(np.asarray(template, np.float32))
    p = torch.from_numpy(np.asarray(pose7, np.float32))
    q = torch.nn.functional.normalize(p[:, :4], dim=1)[:, None, :].expand(-1, t.shape[1], -1)
    qv = q[..., 1:]
    uv = torch.cross(qv, t, dim=2)
    uuv = torch.cross(qv, uv, dim=2)
    return (t + 2 * (q[..., :1] * uv + uuv) + p[:, None, 4:]).numpy()
