#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the BASELINE shapes (HIP-event timing on torch's current stream,
which is the stream the l3d_* launches use).  Prints one line per kernel:
    name  avg_us  achieved  unit  (algorithmic work / time)
Not a product path; used while tuning and for DESIGN.md's tables."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, warm=5, iters=30, reps=3):
    """us per call: the best of `reps` batches of `iters` back-to-back calls between two events.  (A single batch now and
    then contains a one-off 40 ms stall -- allocator growth after a large model ran -- which once showed a 14 us gather as 1.2 ms.)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
        best = t if best is None else min(best, t)
    return best


def main():
    import learning3d_amd.utils as U
    from learning3d_amd.utils import pointnet2_utils as P
    from learning3d_amd.losses.chamfer_distance import ChamferDistance
    from learning3d_amd.models import DGCNN, _fused, PCN
    res = {}
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    B, N, k = 32, 1024, 20
    x = torch.rand((B, N, 3), generator=g).to(dev)
    a = torch.rand((B, N, 3), generator=g).to(dev)
    b = torch.rand((B, N, 3), generator=g).to(dev)
    xt = x.permute(0, 2, 1)
    with torch.no_grad():
        t = timeit(lambda: U.knn(xt, k))
        res["knn_c2"] = (t, B * N * N / t / 1e3, "Gpair/s")
        for Cf in (64, 128):
            xf = torch.randn((B, Cf, N), generator=g).to(dev)
            t = timeit(lambda: U.knn(xf, k))
            res[f"knn_feature_C{Cf}"] = (t, 2.0 * Cf * B * N * N / t / 1e6, "TFLOP/s(fp32-equiv GEMM)")
        cd = ChamferDistance()
        t = timeit(lambda: cd(a, b))
        res["chamfer_c2"] = (t, 2 * B * N * N / t / 1e3, "Gpair/s")
        from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction
        ag, bg = a.clone().requires_grad_(), b.clone().requires_grad_()
        def fwdbwd():
            with torch.enable_grad():
                d1, d2 = ChamferDistanceFunction.apply(ag, bg)
                (d1.sum() + d2.sum()).backward()
        t = timeit(fwdbwd)
        res["chamfer_fwd+bwd_c2"] = (t, 2 * B * N * N / t / 1e3, "Gpair/s(fwd)")
        # the backward kernels alone (raw C-ABI calls): scan of the partner's selections vs the LDS-sorted selection list
        from learning3d_amd._lib import lib as _l, check as _c, ptr as _p, stream_ptr as _s
        for tag, (Bc, Nc, Mc) in (("c2", (B, N, N)), ("c4_B8_16k", (8, 16384, 16384))):
            ca, cb = torch.rand(Bc, Nc, 3, device=dev), torch.rand(Bc, Mc, 3, device=dev)
            d1 = torch.empty(Bc, Nc, device=dev); d2 = torch.empty(Bc, Mc, device=dev)
            i1 = torch.empty(Bc, Nc, dtype=torch.int32, device=dev); i2 = torch.empty(Bc, Mc, dtype=torch.int32, device=dev)
            _c(_l().l3d_chamfer_forward(_p(ca), _p(cb), Bc, Nc, Mc, _p(d1), _p(d2), _p(i1), _p(i2), _s()), "cd")
            g1, g2 = torch.empty_like(ca), torch.empty_like(cb)
            for name, v in (("scan", 0), ("sorted", 2)):
                t = timeit(lambda: _c(_l().l3d_chamfer_backward_variant(_p(ca), _p(cb), Bc, Nc, Mc, _p(d1), _p(d2), _p(i1), _p(i2),
                                                                        _p(g1), _p(g2), v, _s()), "cd bwd"),
                           warm=2, iters=5 if Nc > 4096 else 50)
                res[f"chamfer_bwd_{name}_{tag}"] = (t, (Nc + Mc) * Bc / t, "Mpoint/s")
        # EMD (emd.hip): forward = 20 sweeps + match + costsum, backward = grad1 + grad2; c2 shape and the c4-coarse shape
        for tag, (Be, ne) in (("c2", (32, 1024)), ("c4coarse", (64, 1024))):
            ea, eb = torch.rand(Be, ne, 3, device=dev), torch.rand(Be, ne, 3, device=dev)
            ews = torch.empty(_l().l3d_emd_workspace_bytes(Be, ne, ne), dtype=torch.uint8, device=dev)
            em, ec = torch.empty(Be, ne, ne, device=dev), torch.empty(Be, device=dev)
            eg1, eg2 = torch.empty_like(ea), torch.empty_like(eb)
            t = timeit(lambda: _c(_l().l3d_emd_forward(_p(ea), _p(eb), Be, ne, ne, _p(em), _p(ec), _p(ews), 0, _s()), "emd"), warm=2, iters=10)
            res[f"emd_fwd_{tag}"] = (t, 490.0 * Be * ne * ne / t / 1e6, "T lane-op-eq/s (of 78.6 packed)")
            t = timeit(lambda: _c(_l().l3d_emd_backward(_p(ea), _p(eb), _p(em), Be, ne, ne, _p(eg1), _p(eg2), _s()), "emd bwd"), warm=2, iters=10)
            res[f"emd_bwd_{tag}"] = (t, 2 * 4.0 * Be * ne * ne / t / 1e3, "GB/s (match read twice)")
        net = DGCNN(emb_dims=1024).to(dev).eval()
        idx = U.knn(xt, k)
        packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        t = timeit(lambda: _fused.edgeconv_forward(x, idx, packed, kernel="lds"))
        res["edgeconv_lds_c2"] = (t, B * N * k * 2 * 45440 / t / 1e6, "TFLOP/s")
        t = timeit(lambda: _fused.edgeconv_forward(x, idx, packed, kernel="lds"))
        res["edgeconv_chained_c2"] = (t, B * N * k * 2 * 45440 / t / 1e6, "TFLOP/s")
        t = timeit(lambda: _fused.edgeconv_forward(x, idx, packed, kernel="split"))
        res["edgeconv_bf16x3_c2"] = (t, B * N * k * 2 * 45440 / t / 1e6, "TFLOP/s(fp32-equiv)")
        pooled = _fused.edgeconv_forward(x, idx, packed)
        w5, s5, b5, w5s, w5f = net._conv5_folded()
        t = timeit(lambda: _fused.edgeconv_forward(x, idx, packed, kernel="f16", v2=True))
        res["edgeconv_f16x2_c2"] = (t, B * N * k * 2 * 45440 / t / 1e6, "TFLOP/s(fp32-equiv)")
        t = timeit(lambda: _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True))
        res["edgeconv_f16x2_planes_c2"] = (t, B * N * k * 2 * 45440 / t / 1e6, "TFLOP/s(fp32-equiv)")
        img = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True)
        t = timeit(lambda: _fused.pointwise_conv_f16(img, B, N, w5f, 512, 1024, s5, b5, relu=True))
        res["conv5_f16x2_c2"] = (t, B * N * 2 * 512 * 1024 / t / 1e6, "TFLOP/s(fp32-equiv)")
        t = timeit(lambda: _fused.pointwise_conv(pooled, w5, s5, b5, relu=True, channel_last=True, split=False))
        res["conv5_f32mfma_c2"] = (t, B * N * 2 * 512 * 1024 / t / 1e6, "TFLOP/s")
        t = timeit(lambda: _fused.pointwise_conv(pooled, w5, s5, b5, relu=True, channel_last=True, w_split=w5s))
        res["conv5_bf16x3_c2"] = (t, B * N * 2 * 512 * 1024 / t / 1e6, "TFLOP/s(fp32-equiv)")
        t = timeit(lambda: _fused.split_rows(w5))
        res["split_w5"] = (t, 1024 * 512 * 10 / t / 1e3, "GB/s")
        t = timeit(lambda: net(x))
        res["dgcnn_fwd_c2"] = (t, B / t * 1e6, "clouds/s")
        # c3: SVDHead soft correspondences, B=32, C=512, N=M=1024 (flash-style kernel vs the torch-op composition)
        from learning3d_amd.utils.svd import soft_correspondence
        se = torch.randn((32, 512, 1024), generator=g).to(dev)
        te = torch.randn((32, 512, 1024), generator=g).to(dev)
        tg = (torch.rand((32, 3, 1024), generator=g) - 0.5).to(dev)
        t = timeit(lambda: soft_correspondence(se, te, tg), warm=3, iters=10)
        res["softcorr_c3"] = (t, 2 * 32 * 1024 * 1024 * 512 / t / 1e6, "TFLOP/s(fp32-equiv)")
        def torch_corr():
            sc = torch.matmul(se.transpose(2, 1).contiguous(), te) / (512 ** 0.5)
            sc = torch.softmax(sc, dim=2)
            return torch.matmul(tg, sc.transpose(2, 1).contiguous())
        t = timeit(torch_corr, warm=3, iters=10)
        res["softcorr_c3_torch_ops"] = (t, 2 * 32 * 1024 * 1024 * 512 / t / 1e6, "TFLOP/s")
        # c4 slice: Chamfer 2048 x 16384 (B=8 of 64) -- O(N^2) stress
        a4 = torch.rand((8, 16384, 3), generator=g).to(dev)
        b4 = torch.rand((8, 16384, 3), generator=g).to(dev)
        t = timeit(lambda: cd(a4, b4), warm=20, iters=10)          # long enough for the clocks to settle
        res["chamfer_c4_B8_16k"] = (t, 2 * 8 * 16384 * 16384 / t / 1e3, "Gpair/s")
        pcn = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).to(dev).eval()
        part = (torch.rand((64, 2048, 3), generator=g) - 0.5).to(dev)
        t = timeit(lambda: pcn(part), warm=5, iters=10)
        res["pcn_fwd_c4_B64"] = (t, 64 / t * 1e6, "clouds/s")
        # c5 per-GPU slice: FlowNet3D set-conv grouping, B=32, N=8192, S=1024, r=0.5, K=16
        xyz = torch.clamp(torch.randn((32, 8192, 3), generator=g), -2, 2).to(dev)
        feat = torch.rand((32, 3, 8192), generator=g).to(dev)
        t = timeit(lambda: P.furthest_point_sample(xyz, 1024), warm=1, iters=3)
        res["fps_c5"] = (t, 32 * 1024 * 8192 / t / 1e3, "Gpair/s")
        fps = P.furthest_point_sample(xyz, 1024)
        new_xyz = P.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
        t = timeit(lambda: P.ball_query(0.5, 16, xyz, new_xyz))
        res["ball_query_c5"] = (t, 32 * 1024 * 8192 / t / 1e3, "Gpair/s(max)")
        bidx = P.ball_query(0.5, 16, xyz, new_xyz)
        feat6 = torch.cat([xyz.transpose(1, 2).contiguous(), feat], 1).contiguous()
        t = timeit(lambda: P.grouping_operation(feat6, bidx))
        res["group_c5"] = (t, 27262976 / t / 1e3, "GB/s(alg)")
        t = timeit(lambda: P.knn(64, new_xyz, xyz), warm=1, iters=3)
        res["knn_pair_k64_c5"] = (t, 32 * 1024 * 8192 / t / 1e3, "Gpair/s")
        from learning3d_amd._lib import check, lib, ptr, stream_ptr
        kd = torch.empty((32, 1024, 64), device=dev)
        ki = torch.empty((32, 1024, 64), dtype=torch.int32, device=dev)
        t = timeit(lambda: check(lib().l3d_knn_variant(32, 1024, 8192, 64, ptr(new_xyz), ptr(xyz), ptr(kd), ptr(ki), 1,
                                                       stream_ptr()), "l3d_knn_variant"), warm=1, iters=3)
        res["knn_pair_k64_c5_lane_kernel"] = (t, 32 * 1024 * 8192 / t / 1e3, "Gpair/s")
        t = timeit(lambda: P.three_nn(xyz, new_xyz))          # the last feature-propagation layer: 8192 queries x 1024 candidates
        res["three_nn_c5"] = (t, 32 * 8192 * 1024 / t / 1e3, "Gpair/s")
        l2 = new_xyz[:, :256].contiguous()                   # FlowEmbedding's own shape: 256 x 256 points, nsample 64
        t = timeit(lambda: P.knn(64, l2, l2))
        res["knn_pair_k64_256x256"] = (t, 32 * 256 * 256 / t / 1e3, "Gpair/s")
    for name, (t, v, u) in res.items():
        print(f"{name:22s} {t:10.1f} us   {v:10.2f} {u}")
    print(json.dumps({k: {"us": v[0], "rate": v[1], "unit": v[2]} for k, v in res.items()}))


if __name__ == "__main__":
    main()
