#!/usr/bin/env python3
"""Round-4 additions to the fixtures (the originals are written by make_golden.py and stay as they are): run HERE, where
/root/reference exists, on the imported reference (make_golden.import_reference).

  ppfnet_util.npz        utils/ppfnet_util.py:96-131 query_ball_point(..., itself_indices=...) and :193-243 sample_and_group_multi,
                         the one a-row variant that had no reference-held pin.  The reference's farthest_point_sample starts from
                         torch.randint (:84): both sides are handed the SAME centres -- drawn here with a seeded generator, stored
                         in the fixture, and put in place of the module's farthest_point_sample on either side.
  ptnet_checkpoints.npz  the reference's two other trained PointNet feature extractors (pretrained/exp_ipcrnet, exp_pnlk:
                         best_ptnet_model.t7, use_bn False / True) with their features on a seeded cloud: trained magnitudes for
                         the f16x2 arithmetic beyond config 1's checkpoint.
  prnet_dgcnn_full.npz   models/prnet.py:62-97 DGCNN(emb_dims=512) on 2 clouds of 1024 points with seeded weights: the full-size
                         check of the feature-space kNN chain against the reference instead of against this package's other route.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    tmp, U, Mo, CD = MG.import_reference()
    import learning3d.utils.ppfnet_util as PP
    with torch.no_grad():
        B, N, S, K, r = 2, 300, 40, 12, 0.35
        g = torch.Generator().manual_seed(51)
        xyz = torch.rand((B, N, 3), generator=g) * 2 - 1
        nrm = torch.randn((B, N, 3), generator=g)
        nrm = nrm / nrm.norm(dim=2, keepdim=True)
        fps = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(B)])          # distinct centres per cloud
        new_xyz = PP.index_points(xyz, fps)
        idx_self = PP.query_ball_point(r, K, xyz, new_xyz, fps)
        idx_plain = PP.query_ball_point(r, K, xyz, new_xyz)
        # a radius so small that most balls hold only their own centre: the padding value IS the centre (:125-127)
        idx_tiny = PP.query_ball_point(0.02, K, xyz, new_xyz, fps)
        orig = PP.farthest_point_sample
        PP.farthest_point_sample = lambda x, n: fps
        try:
            out, grouped, fps_ret = PP.sample_and_group_multi(S, r, K, xyz, nrm, returnfps=True)
            nx, npts = PP.sample_and_group(S, r, K, xyz, nrm)
            out_all = PP.sample_and_group_multi(-1, r, K, xyz[:, :64].contiguous(), nrm[:, :64].contiguous())
        finally:
            PP.farthest_point_sample = orig
        assert torch.equal(fps_ret, fps)
        MG.save("ppfnet_util", xyz=xyz, normals=nrm, fps=fps.to(torch.int32), radius=np.float32(r), nsample=K,
                idx_itself=idx_self.to(torch.int32), idx_plain=idx_plain.to(torch.int32), idx_tiny=idx_tiny.to(torch.int32),
                multi_xyz=out["xyz"], multi_dxyz=out["dxyz"], multi_ppf=out["ppf"], multi_grouped=grouped,
                sg_new_xyz=nx, sg_new_points=npts, all_dxyz=out_all["dxyz"], all_ppf=out_all["ppf"])
        # ---- the two trained PointNet feature extractors
        arrs = {}
        x = MG.rand((2, 96, 3), 61, -1.0, 1.0)
        arrs["x"] = x
        for tag, path, use_bn in (("ipcrnet", "pretrained/exp_ipcrnet/models/best_ptnet_model.t7", False),
                                  ("pnlk", "pretrained/exp_pnlk/models/best_ptnet_model.t7", True)):
            ckpt = torch.load(os.path.join(MG.REF, path), map_location="cpu")
            net = Mo.PointNet(emb_dims=1024, use_bn=use_bn).eval()
            net.load_state_dict(ckpt)
            arrs[tag + ".out"] = net(x)
            for k, v in ckpt.items():
                arrs[f"{tag}.w.{k}"] = v
        MG.save("ptnet_checkpoints", **arrs)
        # ---- PRNet's DGCNN at full size (N 1024, emb 512: feature-space kNN at C 64 / 64 / 128, layers 3-4 on the GEMM kernels):
        #      seeded_params() by state_dict key instead of 1.4 MB of stored weights; of the [2, 512, 1024] output every 8th channel
        #      x every 4th point is kept, plus every channel's sum and maximum over the points (all 512 channels pinned)
        from learning3d.models.prnet import DGCNN as PRNetDGCNN
        torch.manual_seed(8)
        pr = MG.seeded_params(PRNetDGCNN(emb_dims=512).eval(), 720)
        xb = MG.rand((2, 3, 1024), 31)
        ob = pr(xb)
        MG.save("prnet_dgcnn_full", x=xb, out_strided=ob[:, ::8, ::4].contiguous(), out_sum=ob.double().sum(-1), out_max=ob.max(-1)[0],
                seed=720)


if __name__ == "__main__":
    main()
