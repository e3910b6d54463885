"""Generate the golden fixtures in this directory by RUNNING THE REAL REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What it does
  * copies /root/reference to a scratch dir named `learning3d` (the reference is a
    namespace package that must be imported under that name; importing it in place
    would write __pycache__ into the read-only tree), stubs the unused `h5py`
    import (utils/transformer.py:4), imports learning3d.{utils,models,losses};
  * loads oracle/_ref/cd_ref*.so = the reference's own Chamfer C++ CPU path
    (losses/cuda/chamfer_distance/chamfer_distance.cpp:59-177), built by
    oracle/build_ref.py from the source where it lies;
  * evaluates each hot-path function on small seeded inputs and stores
    inputs + outputs as compressed .npz next to this script;
  * `make_golden.py <substring> ...` rewrites only the fixtures whose name contains a substring.

The reference ships no tests / KATs; these vectors are what
pins oracle/ (tests/test_oracle_golden.py) and, through it, the HIP kernels.
Nothing here is read at run time on the GPU box except the .npz files.
"""
import glob
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)


def import_reference():
    tmp = tempfile.mkdtemp(prefix="l3d_ref_")
    shutil.copytree(REF, os.path.join(tmp, "learning3d"),
                    ignore=shutil.ignore_patterns("pretrained", "images", "build", "dist", "*.egg-info"))
    sys.dont_write_bytecode = True
    # h5py is not in the image.  The reference imports it at module scope (unused in utils/transformer.py:4; used by
    # data_utils/dataloaders.py:39-44, whose ModelNet40Data() default arguments run at import, :230, :251).  A stand-in whose
    # File() serves the datasets of an .npz written under the .h5 name lets the reference's OWN dataset classes import and run
    # on the synthetic files make_dataset_files() puts where they look (<package>/data/...).
    h5 = types.ModuleType("h5py")

    class _File:
        def __init__(self, name, mode="r"):
            self._z = np.load(name)

        def __getitem__(self, k):
            return self._z[k]

        def close(self):
            self._z.close()
    h5.File = _File
    sys.modules["h5py"] = h5
    make_dataset_files(os.path.join(tmp, "learning3d", "data"))
    # the reference's utils/lib/pointnet2_utils.py imports the compiled `pointnet2_cuda` module, which does not build
    # against torch 2.x: a CPU stand-in made of the oracle's K7-K16 restatements (bit-pinned against the reference's own
    # kernels on the GPU) takes its place, so that FlowNet3D imports and runs (ref_pointnet2_cpu.py)
    import ref_pointnet2_cpu
    sys.modules.setdefault("pointnet2_cuda", ref_pointnet2_cpu.make_module())
    sys.path.insert(0, tmp)
    import learning3d.utils as U          # noqa
    import learning3d.models as Mo        # noqa
    import learning3d.losses.chamfer_distance as CD  # noqa
    return tmp, U, Mo, CD


def make_dataset_files(data_dir):
    """Synthetic stand-ins for the two on-disk datasets, in the reference's layouts: ModelNet40 `ply_data_{train,test}*.h5`
    (here: npz bytes under that name, datasets `data` [n,2048,3], `normal`, `label` [n,1] uint8) + shape_names.txt, and
    FlyingThings3D `TRAIN*/TEST*.npz` (points1/2, color1/2, flow [n,3], valid_mask1 [n] bool)."""
    rng = np.random.default_rng(2024)
    mn = os.path.join(data_dir, "modelnet40_ply_hdf5_2048")
    os.makedirs(mn, exist_ok=True)
    for part, counts in (("train", (5, 4)), ("test", (3,))):
        for i, n in enumerate(counts):
            arrs = dict(data=rng.uniform(-1, 1, (n, 2048, 3)).astype(np.float32), normal=rng.standard_normal((n, 2048, 3)).astype(np.float32),
                        label=rng.integers(0, 40, (n, 1)).astype(np.uint8))
            with open(os.path.join(mn, f"ply_data_{part}{i}.h5"), "wb") as f:
                np.savez(f, **arrs)
    with open(os.path.join(mn, "shape_names.txt"), "w") as f:
        f.write("\n".join(f"class{i}" for i in range(40)) + "\n")
    sf = os.path.join(data_dir, "data_processed_maxcut_35_20k_2k_8192")
    os.makedirs(sf, exist_ok=True)
    for name in ("TRAIN_A_0000_left_0006-0", "TRAIN_A_0001_left_0007-0", "TRAIN_B_0002_right_0008-0", "TEST_A_0000_left_0006-0",
                 "TEST_A_0001_left_0007-0"):
        n = 700
        p1 = (rng.standard_normal((n, 3)) * 5 + np.array([3.0, -2.0, 20.0])).astype(np.float32)
        fl = (rng.standard_normal((n, 3)) * 0.3).astype(np.float32)
        np.savez(os.path.join(sf, name + ".npz"), points1=p1, points2=(p1 + fl + rng.standard_normal((n, 3)) * 0.01).astype(np.float32),
                 color1=rng.uniform(0, 1, (n, 3)).astype(np.float32), color2=rng.uniform(0, 1, (n, 3)).astype(np.float32), flow=fl,
                 valid_mask1=rng.uniform(0, 1, n) > 0.1)
    return mn, sf


def load_cd_ref():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    so = build_ref.main(REF)
    spec = importlib.util.spec_from_file_location("cd_ref", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


from seeded import seeded_params  # noqa: E402  (shared with the tests)


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]     # e.g. `make_golden.py prnet` rewrites only matching files


def save(name, **arrs):
    if ONLY and not any(o in name for o in ONLY):
        return
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def main():
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    tmp, U, Mo, CD = import_reference()
    cd_ref = load_cd_ref()
    from learning3d.utils.model_common_utils import (knn, get_graph_feature, square_distance, index_points,
                                                     farthest_point_sample, knn_point, query_ball_point)
    print("reference imported from", tmp)
    with torch.no_grad():
        # ---- a1 knn (c2 distribution: U(0,1)^3) ------------------------------------------
        x = rand((2, 1024, 3), 0)                                   # [B,N,3]
        idx = knn(x.permute(0, 2, 1), 20)
        save("knn_n1024_k20", xyz=x, idx=idx.to(torch.int16))
        x = rand((3, 200, 3), 1, -1.0, 1.0)
        save("knn_n200_k7", xyz=x, idx=knn(x.permute(0, 2, 1), 7).to(torch.int16),
             idx_plus1=knn(x.permute(0, 2, 1), 7, add_one_to_k=True).to(torch.int16))
        # ---- a2 graph feature -----------------------------------------------------------
        x = rand((2, 96, 3), 2)
        save("graph_feature_n96", xyz=x, feat=get_graph_feature(x.permute(0, 2, 1), k=20, device="cpu").contiguous())
        # ---- a3/a4/a5/a6/a7 -------------------------------------------------------------
        src, dst = rand((2, 64, 3), 3, -1, 1), rand((2, 160, 3), 4, -1, 1)
        save("square_distance", src=src, dst=dst, dist=square_distance(src, dst))
        xyz = rand((2, 512, 3), 5, -1, 1)
        new_xyz = xyz[:, :64, :].contiguous()
        qi, qc = query_ball_point(0.3, 16, xyz, new_xyz, get_cnt=True)
        far = rand((2, 4, 3), 6, 5, 6)                               # empty balls
        qe = query_ball_point(0.3, 16, xyz, far)
        save("query_ball_point", xyz=xyz, new_xyz=new_xyz, idx=qi.to(torch.int32), cnt=qc.to(torch.int32),
             far=far, idx_far=qe.to(torch.int32), radius=np.float32(0.3), nsample=16)
        pts = rand((2, 512, 5), 7)
        save("index_points", points=pts, idx2=qi.to(torch.int32), out2=index_points(pts, qi),
             idx1=qi[:, :, 0].to(torch.int32), out1=index_points(pts, qi[:, :, 0]))
        save("farthest_point_sample", xyz=xyz,
             idx=farthest_point_sample(xyz, 128, start_with_first_point=True).to(torch.int32))
        pos1, pos2 = rand((2, 300, 3), 8), rand((2, 50, 3), 9)
        v, i = knn_point(8, pos1, pos2)
        save("knn_point", pos1=pos1, pos2=pos2, val=v, idx=i.to(torch.int32))
        # ---- a9 chamfer: torch fallback + the reference's C++ nnsearch/backward ----------
        a, b = rand((3, 256, 3), 10), rand((3, 320, 3), 11)
        M = CD.pairwise_distances(a, b)
        d1t, i1t = M.min(2)
        d2t, i2t = M.min(1)
        loss_t = CD.chamfer(a, b)
        d1 = torch.zeros(3, 256); d2 = torch.zeros(3, 320)
        i1 = torch.zeros(3, 256, dtype=torch.int); i2 = torch.zeros(3, 320, dtype=torch.int)
        cd_ref.forward(a, b, d1, d2, i1, i2)
        assert torch.equal(d1, d1t) and torch.equal(d2, d2t), "C++ nnsearch != torch fallback"
        loss_c = (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2.0
        gd1, gd2 = rand((3, 256), 12), rand((3, 320), 13)
        g1 = torch.zeros_like(a); g2 = torch.zeros_like(b)
        cd_ref.backward(a, b, g1, g2, gd1, gd2, i1, i2)
        save("chamfer", xyz1=a, xyz2=b, dist1=d1, dist2=d2, idx1=i1, idx2=i2, loss_fallback=loss_t, loss_ext=loss_c,
             graddist1=gd1, graddist2=gd2, gradxyz1=g1, gradxyz2=g2)
        # ---- a8 DGCNN / PointNet / PCN forwards (eval, randomised BN statistics) ---------
        torch.manual_seed(1)
        net = Mo.DGCNN(emb_dims=64).eval()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.2, 0.2)
        x = rand((2, 128, 3), 14)
        save("dgcnn_emb64", x=x, out=net(x), **{"w." + k: v for k, v in net.state_dict().items()})
        torch.manual_seed(2)
        pn = Mo.PointNet(emb_dims=64, use_bn=True).eval()
        for m in pn.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
        x = rand((2, 100, 3), 15, -1, 1)
        save("pointnet_emb64", x=x, out=pn(x), **{"w." + k: v for k, v in pn.state_dict().items()})
        # ---- a11 SVD head ----------------------------------------------------------------
        from learning3d.utils.svd import SVDHead
        head = SVDHead(emb_dims=32, input_shape="bnc")
        se, te = rand((4, 32, 64), 16, -1, 1), rand((4, 32, 64), 17, -1, 1)
        src = rand((4, 64, 3), 18, -0.5, 0.5)
        ang = rand((4, 3), 19, 0, np.pi / 4)
        Rs = []
        for e in ang:
            cx, cy, cz = torch.cos(e); sx, sy, sz = torch.sin(e)
            Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
            Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            Rs.append(Rz @ Ry @ Rx)
        Rg = torch.stack(Rs)
        tgt = torch.matmul(src, Rg.transpose(1, 2)) + rand((4, 1, 3), 20, -0.5, 0.5)
        # sharpen the soft assignment so H is well conditioned (SURVEY.md section 7 "SVD tolerance")
        te2 = se * 8.0
        R, t = head(se * 8.0, te2, src, tgt)
        save("svd_head", src_emb=se * 8.0, tgt_emb=te2, src=src, tgt=tgt, R=R, t=t)
        Hs = torch.randn(256, 3, 3, generator=torch.Generator().manual_seed(21))
        Rh = []
        for i in range(Hs.shape[0]):
            u, s, v = torch.svd(Hs[i]); r = v @ u.t()
            if torch.det(r) < 0:
                r = (v @ head.reflect) @ u.t()
            Rh.append(r)
        save("svd3x3", H=Hs, R=torch.stack(Rh))
        # ---- config 3: DCP-v2 (DGCNN embed + Transformer pointer + SVD head) ---------------
        torch.manual_seed(5)
        dcp = Mo.DCP(feature_model=Mo.DGCNN(emb_dims=64), cycle=False).eval()
        for m in dcp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
        template = rand((2, 128, 3), 22, -0.5, 0.5)
        source = torch.matmul(template, Rg[:2].transpose(1, 2)) + rand((2, 1, 3), 23, -0.5, 0.5)
        out = dcp(template, source)
        Hs_ = None
        save("dcp_emb64", template=template, source=source, est_R=out["est_R"], est_t=out["est_t"], r=out["r"],
             transformed_source=out["transformed_source"], est_T=out["est_T"],
             **{"w." + k: v for k, v in dcp.state_dict().items()})
        # ---- 8(f) rank 2: PRNet's DGCNN, a k-NN graph in feature space per layer (models/prnet.py:62-97) --
        from learning3d.models.prnet import DGCNN as PRNetDGCNN
        torch.manual_seed(6)
        pr = PRNetDGCNN(emb_dims=64).eval()
        for m in pr.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.2, 0.2)
        x = rand((2, 3, 128), 24)
        save("prnet_dgcnn_emb64", x=x, out=pr(x), **{"w." + k: v for k, v in pr.state_dict().items()})
        # ---- a8 PCN (models/pcn.py:8-153): conv5 is 1029 wide, so emb_dims must be 1024 and the weights are 15 MB.
        #      They are NOT stored: every parameter is overwritten, by state_dict key, with seeded_params() below, and
        #      the tests apply the same function to their model -- only inputs and outputs live in the fixture.
        torch.manual_seed(7)
        pcn = Mo.PCN(emb_dims=1024, num_coarse=64, grid_size=2, detailed_output=True).eval()
        seeded_params(pcn, 700)
        x = rand((2, 300, 3), 25, -0.5, 0.5)
        out = pcn(x)
        save("pcn_seeded", x=x, coarse_output=out["coarse_output"], fine_output=out["fine_output"],
             keys=np.array(sorted(pcn.state_dict().keys())), seed=700)
        # ---- config 4's PCN (num_coarse 1024, grid 4 -> 16384 fine points) on 2 partial clouds of 2048 points --------------
        pcn4 = Mo.PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).eval()
        seeded_params(pcn4, 710)
        x4 = rand((2, 2048, 3), 27, -0.5, 0.5)
        out4 = pcn4(x4)
        save("pcn_seeded_c4", seed_x=27, coarse_output=out4["coarse_output"], fine_output=out4["fine_output"],
             keys=np.array(sorted(pcn4.state_dict().keys())), seed=710)
        # ---- config 1: PointNet classifier with the reference's own trained checkpoint
        #      (pretrained/exp_classifier/models/best_model.t7, examples/test_pointnet.py:98-118, B=8 N=1024) ----------
        ckpt = torch.load(os.path.join(REF, "pretrained", "exp_classifier", "models", "best_model.t7"), map_location="cpu")
        clf = Mo.Classifier(feature_model=Mo.PointNet(emb_dims=1024, use_bn=True)).eval()
        clf.load_state_dict(ckpt)
        x = rand((8, 1024, 3), 0, -1.0, 1.0)                        # U(-1,1), seed 0
        save("classifier_best_model", x=x, logits=clf(x), **{"w." + k: v for k, v in ckpt.items()})
        # ---- utils/pointconv_util.py (north_star names the file): FPS from index 0, smallest-k expanded-distance
        #      kNN (unsorted in the reference -> stored sorted by (distance, index)), density, one density SA layer ----
        import learning3d.utils.pointconv_util as PC
        xyz = rand((2, 512, 3), 26, -1, 1)
        fps = PC.farthest_point_sample(xyz, 64)
        new_xyz = PC.index_points(xyz, fps)
        kidx = PC.knn_point(16, xyz, new_xyz)
        d = PC.square_distance(new_xyz, xyz)
        order = np.lexsort((kidx.numpy(), torch.gather(d, 2, kidx).numpy()), axis=-1)        # by distance, then index
        kidx_sorted = np.take_along_axis(kidx.numpy(), order, axis=-1)
        dens = PC.compute_density(xyz, 0.1)
        torch.manual_seed(8)
        sa = PC.PointConvDensitySetAbstraction(npoint=64, nsample=16, in_channel=3 + 5, mlp=[16, 32], bandwidth=0.1,
                                               group_all=False).eval()
        for m in sa.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
        feats = rand((2, 5, 512), 27)
        sa_xyz, sa_pts = sa(xyz.permute(0, 2, 1), feats)
        save("pointconv_util", xyz=xyz, fps=fps.to(torch.int32), knn_idx_sorted=kidx_sorted.astype(np.int32),
             density=dens, feats=feats, sa_xyz=sa_xyz, sa_points=sa_pts,
             **{"w." + k: v for k, v in sa.state_dict().items()})
        # ---- 8(f) rank 4: DCPTransform (ops/transform_functions.py:271-315), angles / translation fixed by hand so that the
        #      fixture does not depend on numpy's global RNG; apply_transformation is the reference's own scipy path ----------
        from learning3d.ops.transform_functions import DCPTransform
        tf = DCPTransform(angle_range=45, translation_range=1)
        ang = rand((4, 3), 28, 0, np.pi / 4).numpy().astype(np.float64)           # columns: anglex, angley, anglez
        trn = rand((4, 3), 29, -1, 1).numpy().astype(np.float64)
        tmpl = rand((4, 200, 3), 30, -0.5, 0.5)
        srcs, igts = [], []
        for i in range(4):
            tf.anglex, tf.angley, tf.anglez = ang[i]
            tf.translation = trn[i]
            srcs.append(torch.from_numpy(tf.apply_transformation(tmpl[i].numpy())).float())
            igts.append(tf.igt)
        save("dcp_transform", template=tmpl, anglex=ang[:, 0], angley=ang[:, 1], anglez=ang[:, 2], translation=trn,
             source=torch.stack(srcs), igt=torch.stack(igts))
        # ---- 8(f) rank 4, the other pose generators: PNLKTransform / RPMNetTransform (twist -> se3.exp, :109-192) and
        #      PCRNetTransform (quaternion + translation, :194-269), transforms fixed by hand (the classes draw with torch /
        #      numpy global RNGs) and applied by the reference's own apply_transform / __call__ ------------------------------
        from learning3d.ops.transform_functions import PNLKTransform, RPMNetTransform, PCRNetTransform
        twist = torch.cat([rand((5, 6), 33, -1, 1), torch.tensor([[1e-3, -2e-3, 5e-4, 0.3, -0.2, 0.1]])])   # last: Taylor branch of the sinc helpers
        tmpl = rand((6, 150, 3), 34, -0.5, 0.5)
        nrm = torch.nn.functional.normalize(rand((6, 150, 3), 35, -1, 1), dim=2)
        src, igt, gt, src6 = [], [], [], []
        for i in range(6):
            t1 = PNLKTransform(mag=1)
            src.append(t1.apply_transform(tmpl[i], twist[i:i + 1])); igt.append(t1.igt); gt.append(t1.gt)
            t2 = RPMNetTransform(mag=1)
            src6.append(t2.apply_transform(torch.cat([tmpl[i], nrm[i]], dim=1), twist[i:i + 1]))
        pose = torch.cat([rand((6, 4), 36, -1, 1), rand((6, 3), 37, -1, 1)], dim=1)                     # un-normalised quaternion + t
        psrc = []
        for i in range(6):
            t3 = PCRNetTransform(1, angle_range=45, translation_range=1)
            t3.transformations = [pose[i:i + 1]]
            t3.index = 0
            psrc.append(t3(tmpl[i]))
        save("pose_transforms", template=tmpl, normals=nrm, twist=twist, source=torch.stack(src), igt=torch.stack(igt),
             gt=torch.stack(gt), source6=torch.stack(src6), pose7=pose, pcr_source=torch.stack(psrc))
        # ---- CurveNet LPFA (utils/curvenet_util.py:229-291): kNN on xyz with add_one_to_k, grouping, both variants ------
        from learning3d.utils.curvenet_util import LPFA
        torch.manual_seed(9)
        xyz = rand((2, 3, 200), 31)
        feats = rand((2, 16, 200), 32, -1, 1)
        outs = {}
        for name, initial in (("init", True), ("deep", False)):
            m = LPFA(9 if initial else 16, 24, k=12, mlp_num=1 if initial else 2, initial=initial).eval()
            m.device = torch.device("cpu")
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
            outs["out_" + name] = m(xyz if initial else feats, xyz)            # models/curvenet.py:62 calls lpfa(xyz, xyz)
            outs.update({f"w_{name}." + k: v for k, v in m.state_dict().items()})
        save("lpfa", xyz=xyz, feats=feats, **outs)
        # ---- FlowNet3D as a model (models/flownet3d.py:73-328): the reference's own module + its own pointnet2_utils.py
        #      wrappers, running on the CPU stand-in for pointnet2_cuda; seeded weights by state_dict key --------------
        import learning3d.utils as U2
        assert hasattr(U2, "pointnet2_utils"), "the reference's pointnet2_utils did not import"
        from learning3d.models.flownet3d import FlowNet3D, PointNetSetAbstraction
        cuda_types = (torch.cuda.IntTensor, torch.cuda.FloatTensor)
        torch.cuda.IntTensor, torch.cuda.FloatTensor = torch.IntTensor, torch.FloatTensor   # pointnet2_utils.py:25-28 allocates with these
        try:
            fn = FlowNet3D().eval()
            seeded_params(fn, 900)
            g = torch.Generator().manual_seed(90)
            pc1 = torch.clamp(torch.randn((2, 3, 2048), generator=g), -2, 2)
            pc2 = (pc1 + 0.05 * torch.randn((2, 3, 2048), generator=g)).contiguous()
            f1, f2 = torch.rand((2, 3, 2048), generator=g), torch.rand((2, 3, 2048), generator=g)
            l1_pc1, l1_f1 = fn.sa1(pc1, f1)
            l2_pc1, l2_f1 = fn.sa2(l1_pc1, l1_f1)
            l1_pc2, l1_f2 = fn.sa1(pc2, f2)
            l2_pc2, l2_f2 = fn.sa2(l1_pc2, l1_f2)
            _, l2_f1_new = fn.fe_layer(l2_pc1, l2_pc2, l2_f1, l2_f2)
            sf = fn(pc1, pc2, f1, f2)
            save("flownet3d_seeded", pc1=pc1, pc2=pc2, f1=f1, f2=f2, sf=sf, l1_pc1=l1_pc1, l1_feature1=l1_f1,
                 l2_feature1=l2_f1, l2_feature1_new=l2_f1_new, keys=np.array(sorted(fn.state_dict().keys())), seed=900)
            # config 5's layer at its own shape on 4 clouds (sa1: npoint 1024 of N 8192, r 0.5, K 16, mlp 32/32/64)
            sa = PointNetSetAbstraction(npoint=1024, radius=0.5, nsample=16, in_channel=3, mlp=[32, 32, 64], group_all=False).eval()
            seeded_params(sa, 910)
            g = torch.Generator().manual_seed(0)
            xyz = torch.clamp(torch.randn((4, 3, 8192), generator=g), -2, 2)
            feat = torch.rand((4, 3, 8192), generator=torch.Generator().manual_seed(1))
            new_xyz, new_feat = sa(xyz, feat)
            save("flownet3d_sa1_c5", seed_xyz=0, seed_feat=1, new_xyz=new_xyz, new_feat=new_feat.to(torch.float32), seed=910,
                 keys=np.array(sorted(sa.state_dict().keys())))
        finally:
            torch.cuda.IntTensor, torch.cuda.FloatTensor = cuda_types
    # ---- f4, the on-disk half: the reference's own dataset classes (data_utils/dataloaders.py:184-247, :364-435) on the synthetic
    #      files; the fixture carries the files' arrays so that the tests can rebuild them on the GPU box ------------------------
    import learning3d.data_utils.dataloaders as DL
    data_dir = os.path.join(tmp, "learning3d", "data")
    mn, sf = os.path.join(data_dir, "modelnet40_ply_hdf5_2048"), os.path.join(data_dir, "data_processed_maxcut_35_20k_2k_8192")
    out = {}
    for fn in sorted(glob.glob(os.path.join(mn, "ply_data_*.h5"))):
        z = np.load(fn)
        for k in ("data", "normal", "label"):
            out[f"mn.{os.path.basename(fn)[:-3]}.{k}"] = z[k][:, :256] if k != "label" else z[k]     # the tests use num_points <= 256
    for part, train in (("train", True), ("test", False)):
        ds = DL.ModelNet40Data(train=train, num_points=128, download=False, randomize_data=False)
        # the reference concatenates in glob order (file-system dependent): record it
        out[f"mn.{part}.order"] = np.array([os.path.basename(f)[:-3] for f in glob.glob(os.path.join(mn, f"ply_data_{part}*.h5"))])
        pts, lab = zip(*[ds[i] for i in range(len(ds))])
        out[f"mn.{part}.points"], out[f"mn.{part}.labels"] = torch.stack(pts).numpy(), torch.stack(lab).numpy()
    ds = DL.ModelNet40Data(train=True, num_points=128, download=False, randomize_data=True, use_normals=True)
    np.random.seed(77)
    out["mn.rand.points"] = torch.stack([ds[i][0] for i in (0, 3, 8)]).numpy()
    cls = DL.ClassificationData(DL.ModelNet40Data(train=False, num_points=64, download=False))
    out["mn.cls.points"], out["mn.cls.label"] = cls[1][0].numpy(), cls[1][1].numpy()
    out["mn.cls.shape"] = np.array(str(cls.get_shape(int(cls[1][1]))))
    for fn in sorted(glob.glob(os.path.join(sf, "*.npz"))):
        z = np.load(fn)
        for k in z.files:
            out[f"sf.{os.path.basename(fn)[:-4]}.{k}"] = z[k]
    for part in ("train", "test"):
        ds = DL.SceneflowDataset(npoints=256, root=sf, partition=part)
        out[f"sf.{part}.order"] = np.array([os.path.basename(f)[:-4] for f in ds.datapath])
        np.random.seed(91)
        for i in range(len(ds)):
            item = ds[i]
            for name, v in zip(("pos1", "pos2", "color1", "color2", "flow", "mask1"), item):
                out[f"sf.{part}.{i}.{name}"] = np.array(v)
    save("datasets", **out)
    shutil.rmtree(tmp, ignore_errors=True)
    print("done")


if __name__ == "__main__":
    main()
