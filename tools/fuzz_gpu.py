"""Randomised shape sweep of the fused kernels against torch / fp64 evaluations (run on a GPU box).
Complements tests/test_gpu_parity.py (fixed shapes); prints the worst error per kernel family."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learning3d_amd.models import DGCNN, _fused
import learning3d_amd.utils as U
from learning3d_amd.utils.svd import soft_correspondence
from learning3d_amd._lib import lib, check, ptr, stream_ptr
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
worst = {}
def rec(name, got, want, rtol, atol):
    err = np.abs(got - want) - rtol * np.abs(want)
    worst[name] = max(worst.get(name, -1), float(err.max()))
    assert err.max() <= atol, (name, err.max())
torch.manual_seed(0)
net = DGCNN(emb_dims=64).cuda().eval()
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
with torch.no_grad():
    for it in range(12):                                            # EdgeConv: all three kernels agree
        B, N, k = int(rng.integers(1, 5)), int(rng.integers(21, 700)), int(rng.integers(1, 21))
        x = dev(rng.uniform(0, 1, (B, N, 3)).astype(np.float32))
        idx = U.knn(x.permute(0, 2, 1), k)
        packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        a = _fused.edgeconv_forward(x, idx, packed, kernel="lds").cpu().numpy()
        s = _fused.edgeconv_forward(x, idx, packed, kernel="split").cpu().numpy()
        c = _fused.edgeconv_forward(x, idx, packed, kernel="lds").cpu().numpy()
        rec("edgeconv split vs lds", s, a, 1e-5, 2e-6); rec("edgeconv chained vs lds", c, a, 1e-5, 2e-6)
    for it in range(12):                                            # 1x1 conv: bf16x3 vs fp32-MFMA vs fp64
        B = int(rng.integers(1, 4)); Cin = 16 * int(rng.integers(2, 40)); Cout = 256 * int(rng.integers(1, 4)); N = 128 * int(rng.integers(1, 9))
        x = rng.standard_normal((B, Cin, N)).astype(np.float32); w = (rng.standard_normal((Cout, Cin)) / math.sqrt(Cin)).astype(np.float32)
        sh = rng.standard_normal((B, Cout)).astype(np.float32)
        want = np.maximum(np.einsum("oc,bcn->bon", w.astype(np.float64), x) + sh[:, :, None], 0)
        for cl in (False, True):
            xin = dev(x.transpose(0, 2, 1)) if cl else dev(x)
            got = _fused.pointwise_conv(xin, dev(w), None, dev(sh), relu=True, channel_last=cl, split=True).cpu().numpy()
            rec("conv_split vs fp64", got, want, 1e-5, 5e-6)
        K = int(rng.choice([8, 16, 32, 64]))
        if N % K == 0:
            full = _fused.pointwise_conv(dev(x), dev(w), None, dev(sh), relu=True, split=False)
            pooled = _fused.pointwise_conv_maxpool(dev(x), dev(w), None, dev(sh), True, K)
            # the pooled epilogue exists in both GEMM arithmetics (bf16x3 when N % 256 == 0): fp32-level agreement
            rec("conv maxpool", pooled.cpu().numpy(), full.view(B, Cout, N // K, K).max(-1)[0].cpu().numpy(), 1e-5, 5e-6)
    for it in range(10):                                            # soft correspondence + attention, ragged
        B = int(rng.integers(1, 3)); C = 16 * int(rng.integers(2, 20)); N = int(rng.integers(1, 600)); M = int(rng.integers(1, 600))
        q = rng.standard_normal((B, C, N)).astype(np.float32); k_ = rng.standard_normal((B, C, M)).astype(np.float32)
        v = rng.uniform(-1, 1, (B, 3, M)).astype(np.float32)
        s = np.einsum("bcn,bcm->bnm", q.astype(np.float64), k_.astype(np.float64)) / math.sqrt(C)
        s = np.exp(s - s.max(2, keepdims=True)); s /= s.sum(2, keepdims=True)
        rec("softcorr vs fp64", soft_correspondence(dev(q), dev(k_), dev(v)).cpu().numpy(), np.einsum("bdm,bnm->bdn", v.astype(np.float64), s), 1e-5, 4e-6)
        H = int(rng.choice([1, 2, 4])); D = int(rng.choice([32, 64, 128]))
        qa = rng.standard_normal((B, H, D, N)).astype(np.float32); ka = rng.standard_normal((B, H, D, M)).astype(np.float32)
        va = rng.standard_normal((B, H, D, M)).astype(np.float32)
        s = np.einsum("bhdn,bhdm->bhnm", qa.astype(np.float64), ka.astype(np.float64)) / math.sqrt(D)
        s = np.exp(s - s.max(-1, keepdims=True)); s /= s.sum(-1, keepdims=True)
        want = np.einsum("bhdm,bhnm->bhdn", va.astype(np.float64), s)
        qd, kd, vd = dev(qa.reshape(B, H * D, N)), dev(ka.reshape(B, H * D, M)), dev(va.reshape(B, H * D, M))
        out = torch.empty_like(qd)
        check(lib().l3d_attention_forward_strided(ptr(qd), ptr(kd), ptr(vd), B, H, D, N, M, H * D * N, H * D * M, H * D * M, 1 / math.sqrt(D), ptr(out), stream_ptr()), "att")
        rec("attention vs fp64", out.cpu().numpy().reshape(B, H, D, N), want, 1e-5, 4e-6)
    for it in range(10):                                            # Chamfer: packed kernel == per-candidate kernel, bit for bit
        B = int(rng.integers(1, 4)); N = int(rng.integers(1, 3000)); M = int(rng.integers(1, 3000))
        a = dev(rng.uniform(0, 1, (B, N, 3)).astype(np.float32)); b_ = dev(np.round(rng.uniform(0, 1, (B, M, 3)) * 8).astype(np.float32) / 8)   # ties
        outs = []
        for m_ in (0, 2):
            d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
            i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
            check(lib().l3d_chamfer_forward_variant(ptr(a), ptr(b_), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), m_, stream_ptr()), "cd")
            outs.append([t.cpu().numpy() for t in (d1, d2, i1, i2)])
        for x_, y_ in zip(*outs):
            assert np.array_equal(x_, y_), ("chamfer packed", B, N, M)
    worst["chamfer packed == scalar"] = 0.0
    for it in range(8):                                             # feature-space kNN vs exact fp64 distances
        B = int(rng.integers(1, 3)); C = 32 * int(rng.integers(1, 9)); N = int(rng.integers(21, 1500)); k = int(rng.integers(1, 21))
        x = rng.standard_normal((B, C, N)).astype(np.float32)
        idx = U.knn(dev(x), k).cpu().numpy()
        xd = x.astype(np.float64); sq = (xd ** 2).sum(1)
        d = sq[:, :, None] + sq[:, None, :] - 2 * np.einsum("bci,bcj->bij", xd, xd)
        kth = np.sort(d, axis=-1)[:, :, k - 1]
        got = np.take_along_axis(d, idx, axis=-1)
        rec("feature knn: k-th bound", got.max(-1), kth, 0, 4e-6 * sq.max())
        assert np.all(np.diff(got, axis=-1) >= -4e-6 * sq.max()) and np.all(idx[:, :, 0] == np.arange(N)[None])
    from learning3d_amd.utils import pointnet2_utils as P
    for it in range(6):                                             # deterministic scatter-add vs fp64 index_add
        B = int(rng.integers(1, 4)); C = int(rng.integers(1, 70)); T = int(rng.integers(1, 900)); E = int(rng.integers(1, 5000))
        src = rng.standard_normal((B, C, E)).astype(np.float32); idx = rng.integers(0, T, (B, E)).astype(np.int32)
        got = P._scatter_add_det(dev(src), dev(idx), None, T, 1).cpu().numpy()
        ref = np.zeros((B, C, T))
        for b in range(B):
            np.add.at(ref[b], (slice(None), idx[b]), src[b].astype(np.float64))
        rec("scatter_add_det vs fp64", got, ref, 1e-5, 2e-5)
# ---- round 3's kernels ------------------------------------------------------------------------------------------------------
with torch.no_grad():
    from learning3d_amd.models import _train
    for it in range(10):                                            # Chamfer backward: sorted selection list == scan, bit for bit
        B = int(rng.integers(1, 4)); N = int(rng.integers(1, 3000)); M = int(rng.integers(1, 3000))
        a = dev(rng.uniform(0, 1, (B, N, 3)).astype(np.float32)); b_ = dev(np.round(rng.uniform(0, 1, (B, M, 3)) * 6).astype(np.float32) / 6)
        d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
        i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
        check(lib().l3d_chamfer_forward(ptr(a), ptr(b_), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd")
        g1 = dev(rng.standard_normal((B, N)).astype(np.float32)); g2 = dev(rng.standard_normal((B, M)).astype(np.float32))
        outs = []
        for v_ in (0, 2):
            x1 = torch.full((B, N, 3), float("nan"), device="cuda"); x2 = torch.full((B, M, 3), float("nan"), device="cuda")
            check(lib().l3d_chamfer_backward_variant(ptr(a), ptr(b_), B, N, M, ptr(g1), ptr(g2), ptr(i1), ptr(i2), ptr(x1), ptr(x2), v_,
                                                     stream_ptr()), "cd bwd")
            outs.append((x1.cpu().numpy(), x2.cpu().numpy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), ("chamfer bwd", B, N, M)
    worst["chamfer bwd sorted == scan"] = 0.0
    for it in range(10):                                            # max over the last axis (+ backward) == torch
        shape = tuple(int(v) for v in rng.integers(1, 40, int(rng.integers(1, 4)))) + (int(rng.integers(1, 257)),)
        x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda().relu()
        with torch.enable_grad():
            a_, b_ = x.clone().requires_grad_(), x.clone().requires_grad_()
            va = _train.max_over_last(a_); vb = b_.max(dim=-1, keepdim=True)[0]
            w_ = torch.randn_like(va)
            (va * w_).sum().backward(); (vb * w_).sum().backward()
        assert torch.equal(va, vb) and torch.equal(a_.grad, b_.grad), ("max_last", shape)
    worst["max_last == torch.max"] = 0.0
    for it in range(8):                                             # linear over a handful of rows vs fp64
        R = int(rng.integers(1, 300)); Cin = 256 * int(rng.integers(1, 5)); Cout = int(rng.integers(1, 700))
        lin = torch.nn.Linear(Cin, Cout).cuda()
        x = dev(rng.standard_normal((R, Cin)).astype(np.float32))
        want = (x.double() @ lin.weight.double().t() + lin.bias.double()).clamp_min(0).cpu().numpy()
        rec("linear_rows vs fp64", _fused.linear_rows(x, lin, True).cpu().numpy(), want, 1e-5, 1e-5 * max(1.0, float(np.abs(want).max())))
    for it in range(8):                                             # channel-first LayerNorm vs fp64; residual epilogue == conv + add
        B = int(rng.integers(1, 4)); C = int(rng.choice([128, 256, 512])); N = int(rng.integers(1, 700))
        x = (rng.standard_normal((B, C, N)) * rng.uniform(0.01, 30.0) + rng.uniform(-5, 5)).astype(np.float32)
        a = rng.uniform(0.5, 1.5, C).astype(np.float32); b_ = rng.uniform(-0.5, 0.5, C).astype(np.float32)
        x64 = x.astype(np.float64)
        want = a[None, :, None] * (x64 - x64.mean(1, keepdims=True)) / (x64.std(1, ddof=1, keepdims=True) + 1e-6) + b_[None, :, None]
        tx, ta, tb = dev(x), dev(a), dev(b_)
        y = torch.empty((B, C, N), device="cuda")
        check(lib().l3d_layernorm_planes_cf(ptr(tx), ptr(ta), ptr(tb), 1e-6, B, C, N, ptr(y), None, 0, stream_ptr()), "ln cf")
        rec("layernorm cf vs fp64", y.cpu().numpy(), want, 2e-6, 4e-6)
        C1 = 256 * int(rng.integers(1, 3)); N2 = 256 * int(rng.integers(1, 4)); C0 = 16 * int(rng.integers(2, 30))
        xr = rng.standard_normal((B, N2, C0)).astype(np.float32)
        ximg = _fused.split_rows_f16(dev(xr)); wimg = _fused.split_weights_f16(dev((rng.standard_normal((C1, C0)) / math.sqrt(C0)).astype(np.float32)))
        res = dev(rng.standard_normal((B, C1, N2)).astype(np.float32)); sh = dev(rng.standard_normal(C1).astype(np.float32))
        plain = _fused.pointwise_conv_f16(ximg, B, N2, wimg, C0, C1, None, sh)
        assert torch.equal(_fused.pointwise_conv_f16(ximg, B, N2, wimg, C0, C1, None, sh, residual=res), res + plain), ("residual", B, C0, C1, N2)
    worst["conv residual == conv + add"] = 0.0
    net1k = DGCNN(emb_dims=1024).cuda().eval()
    for it in range(8):                                             # EdgeConv: the two-plane persistent kernel vs the fp32-MFMA one; kNN kernels agree
        B = int(rng.integers(1, 5)); N = int(rng.integers(21, 1300)); k = 20
        x = dev(rng.uniform(0, 1, (B, N, 3)).astype(np.float32))
        idx = U.knn(x.permute(0, 2, 1), k)
        packed = net1k._packed.get([net1k.conv1, net1k.conv2, net1k.conv3, net1k.conv4], [net1k.bn1, net1k.bn2, net1k.bn3, net1k.bn4], x.device)
        ref = _fused.edgeconv_forward(x, idx, packed, kernel="lds").cpu().numpy()
        got = _fused.edgeconv_forward(x, idx, packed, kernel="f16", v2=net1k._packed.v2_ok).cpu().numpy()
        rec("edgeconv f16b vs fp32 MFMA", got, ref, 1e-5, 2e-5 * float(np.abs(ref).max()))
        i0 = torch.empty((B, N, k), dtype=torch.int64, device="cuda"); i1_ = torch.empty_like(i0)
        xc = x.contiguous()
        if lib().l3d_knn_graph_variant(ptr(xc), B, N, k, ptr(i1_), 2, stream_ptr()) == 0:
            check(lib().l3d_knn_graph_variant(ptr(xc), B, N, k, ptr(i0), 1, stream_ptr()), "knn")
            d0 = (xc[:, :, None, :] - torch.gather(xc[:, None].expand(B, N, N, 3), 2, i0[..., None].expand(B, N, k, 3))).square().sum(-1)
            d1_ = (xc[:, :, None, :] - torch.gather(xc[:, None].expand(B, N, N, 3), 2, i1_[..., None].expand(B, N, k, 3))).square().sum(-1)
            assert torch.equal(i0, i1_) or torch.allclose(d0, d1_, rtol=0, atol=1e-6), ("knn mfma vs insertion", B, N)
    # ---- round 4: the strided batched GEMM, the cell-list ball query and the fused set-abstraction kernel
    from learning3d_amd.models import _rows, PointNetSetAbstraction
    from learning3d_amd.utils import pointnet2_utils as P2
    for it in range(20):                                            # l3d_bmm_f32: random shapes, strides, transposes, head splits
        nb1, nb2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        M, N_, K = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 400))
        def operand(r, c):
            big = dev(rng.standard_normal((nb1, nb2, r + 5, c + 3)).astype(np.float32))
            mode = int(rng.integers(0, 4))
            if mode == 0:
                return big[:, :, :r, :c].contiguous()
            if mode == 1:
                return big[:, :, 2:2 + r, 1:1 + c]                                      # a sliced view
            if mode == 2:
                return dev(rng.standard_normal((nb1, nb2, c, r)).astype(np.float32)).transpose(-1, -2)   # transposed
            return dev(rng.standard_normal((nb1, r, nb2, c)).astype(np.float32)).transpose(1, 2)         # head-split layout
        A_, B_ = operand(M, K), operand(K, N_)
        parts = int(rng.choice([1, 1, 2, 5]))
        got = _rows.bmm(A_, B_, alpha=0.75, parts=parts).double().cpu().numpy()
        want = 0.75 * np.matmul(A_.double().cpu().numpy(), B_.double().cpu().numpy())
        bound = 0.75 * np.matmul(np.abs(A_.double().cpu().numpy()), np.abs(B_.double().cpu().numpy())) * (K + parts) * 2.0 ** -24 + 1e-30
        worst["bmm vs fp64 / bound"] = max(worst.get("bmm vs fp64 / bound", 0.0), float((np.abs(got - want) / bound).max()))
        assert (np.abs(got - want) <= bound).all(), ("bmm", nb1, nb2, M, N_, K, parts)
    for it in range(10):                                            # ball query: cell list == scanning kernel
        B = int(rng.integers(1, 4)); N = int(rng.integers(2048, 9000)); S = int(rng.integers(1, 700)); K = int(rng.integers(1, 65))
        scale = float(rng.choice([0.3, 1.0, 5.0])); r = float(rng.uniform(0.02, 0.8)) * scale
        xyz = (rng.standard_normal((B, N, 3)) * scale).astype(np.float32)
        if it % 3 == 0:
            xyz = np.clip(xyz, -0.8 * scale, 0.8 * scale)
        new = (rng.standard_normal((B, S, 3)) * scale * 1.3).astype(np.float32)
        xd, nd = dev(xyz), dev(new)
        P2.BALL_QUERY_CELLS = True; a = P2.ball_query(r, K, xd, nd)
        P2.BALL_QUERY_CELLS = False; b_ = P2.ball_query(r, K, xd, nd)
        P2.BALL_QUERY_CELLS = True
        assert torch.equal(a, b_), ("ball query cells", B, N, S, K, r, int((a != b_).sum()))
    worst["ball query cells == scan"] = 0.0
    for it in range(8):                                             # fused set-abstraction kernel vs group + conv launches
        B = int(rng.integers(1, 3)); N = int(rng.integers(300, 3000)); S = int(rng.integers(1, 200)); K = int(rng.choice([8, 16, 32, 64]))
        D = int(rng.choice([0, 1, 3, 5, 13])); widths = [32, 32, 64] if rng.integers(0, 2) else [64, 64, 128]
        sa = PointNetSetAbstraction(npoint=S, radius=float(rng.uniform(0.2, 0.9)), nsample=K, in_channel=D, mlp=list(widths), group_all=False).cuda().eval()
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
        xyz = dev(rng.standard_normal((B, 3, N)).astype(np.float32)); feat = dev(rng.uniform(-1, 1, (B, D, N)).astype(np.float32)) if D else None
        _, f1 = sa(xyz, feat)
        _fused.SA_FUSED = False
        _, f0 = sa(xyz, feat)
        _fused.SA_FUSED = True
        rec("sa fused vs layer kernels", f1.cpu().numpy(), f0.cpu().numpy(), 1e-4, 1e-5)
    _fused.check_range(sync=True)
for k_, v_ in worst.items():
    print(f"{k_:28s} worst (|err| - rtol|want|) = {v_:.3e}")
print("fuzz OK")
