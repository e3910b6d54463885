"""SURVEY.md 8(f) rank 4, the on-disk half: learning3d_amd.data_utils.disk_feed against the reference's OWN dataset classes
(data_utils/dataloaders.py:184-247 ModelNet40Data / ClassificationData, :364-435 SceneflowDataset), which make_golden.py ran
on synthetic files in the reference's layouts (tests/golden/datasets.npz carries the files' arrays and the classes' outputs).
CPU part: the drop-in Dataset classes, value for value (same numpy RNG call sequence).  GPU part (-m gpu): the resident feeds
(dataset in HBM, whole batches from l3d_sceneflow_batch / device gathers), bit-identical to the reference's items."""
import os

import numpy as np
import pytest
import torch


def _write_files(g, root):
    mn = os.path.join(root, "modelnet40_ply_hdf5_2048")
    sf = os.path.join(root, "data_processed_maxcut_35_20k_2k_8192")
    os.makedirs(mn), os.makedirs(sf)
    stems = sorted({k.split(".")[1] for k in g if k.startswith("mn.ply_data_")})
    for st in stems:
        np.savez(os.path.join(mn, st + ".npz"), **{k: g[f"mn.{st}.{k}"] for k in ("data", "normal", "label")})
    with open(os.path.join(mn, "shape_names.txt"), "w") as f:
        f.write("\n".join(f"class{i}" for i in range(40)) + "\n")
    for st in sorted({k.split(".")[1] for k in g if k.startswith("sf.T")}):
        np.savez(os.path.join(sf, st + ".npz"), **{k: g[f"sf.{st}.{k}"] for k in ("points1", "points2", "color1", "color2", "flow", "valid_mask1")})
    return mn, sf


def _ref_rows(g, part):
    """cloud order of the reference's concatenation (its glob order) -> row offsets into OUR sorted concatenation"""
    sizes = {st: g[f"mn.{st}.label"].shape[0] for st in sorted({k.split(".")[1] for k in g if k.startswith(f"mn.ply_data_{part}")})}
    start, off = {}, 0
    for st in sorted(sizes):
        start[st] = off
        off += sizes[st]
    return np.concatenate([np.arange(start[str(st)], start[str(st)] + sizes[str(st)]) for st in g[f"mn.{part}.order"]])


def test_modelnet40_and_sceneflow_dropins_match_the_reference_classes(golden, tmp_path):
    from learning3d_amd.data_utils import disk_feed as D
    g = golden("datasets")
    mn, sf = _write_files(g, str(tmp_path))
    for part, train in (("train", True), ("test", False)):
        ds = D.ModelNet40Data(train=train, num_points=128, root=str(tmp_path))
        rows = _ref_rows(g, part)
        assert len(ds) == len(rows)
        for j, i in enumerate(rows):
            pts, lab = ds[int(i)]
            assert pts.dtype == torch.float32 and lab.dtype == torch.int64 and lab.shape == (1,)
            assert np.array_equal(pts.numpy(), g[f"mn.{part}.points"][j]) and np.array_equal(lab.numpy(), g[f"mn.{part}.labels"][j])
    # randomize_data + use_normals under the same numpy seed (ModelNet40Data.randomize, :214-216)
    rows = _ref_rows(g, "train")
    ds = D.ModelNet40Data(train=True, num_points=128, randomize_data=True, use_normals=True, root=str(tmp_path))
    np.random.seed(77)
    got = torch.stack([ds[int(rows[i])][0] for i in (0, 3, 8)]).numpy()
    assert got.shape == (3, 128, 6) and np.array_equal(got, g["mn.rand.points"])
    cls = D.ClassificationData(D.ModelNet40Data(train=False, num_points=64, root=str(tmp_path)))
    i = int(_ref_rows(g, "test")[1])
    assert np.array_equal(cls[i][0].numpy(), g["mn.cls.points"]) and np.array_equal(cls[i][1].numpy(), g["mn.cls.label"])
    assert str(cls.get_shape(int(cls[i][1]))) == str(g["mn.cls.shape"])
    with pytest.raises(FileNotFoundError):
        D.ModelNet40Data(root=os.path.join(str(tmp_path), "nowhere"))
    # scene flow: same RNG call sequence -> same samples -> same items (first access; the reference's test partition
    # modifies its cache in place on later accesses, dataloaders.py:427-429 on views of the cached arrays)
    for part in ("train", "test"):
        ds = D.SceneflowDataset(npoints=256, root=sf, partition=part)
        order = [str(s) for s in g[f"sf.{part}.order"]]
        mine = [os.path.basename(p)[:-4] for p in ds.datapath]
        assert sorted(order) == mine
        np.random.seed(91)
        for i, st in enumerate(order):                     # the reference visited its files in ITS glob order
            item = ds[mine.index(st)]
            for name, v in zip(("pos1", "pos2", "color1", "color2", "flow", "mask1"), item):
                want = g[f"sf.{part}.{i}.{name}"]
                assert v.dtype == want.dtype and np.array_equal(v, want), (part, i, name)


def test_numpy_axis0_mean_is_sequential_fp32_then_fp64_divide():
    """What l3d_sceneflow_batch replays (feed.hip): np.mean(pos1, 0) of a float32 [S,3] array = rows added one after the other
    in fp32, the quotient formed in fp64 and rounded once."""
    rng = np.random.default_rng(3)
    for n in (7, 256, 2048, 8192):
        a = (rng.standard_normal((n, 3)) * 5 + np.array([3.0, -2.0, 20.0])).astype(np.float32)
        acc = np.zeros(3, np.float32)
        for i in range(n):
            acc = (acc + a[i]).astype(np.float32)
        assert np.array_equal(np.mean(a, 0), (acc.astype(np.float64) / float(n)).astype(np.float32))


@pytest.mark.gpu
def test_resident_feeds_are_bit_identical_to_the_reference_items(golden, tmp_path):
    from learning3d_amd.data_utils import disk_feed as D
    g = golden("datasets")
    mn, sf = _write_files(g, str(tmp_path))
    # ModelNet40 from HBM: explicit cloud indices
    for part, train in (("train", True), ("test", False)):
        feed = D.ResidentModelNet40(train=train, num_points=128, batch_size=4, root=str(tmp_path))
        rows = torch.as_tensor(_ref_rows(g, part)).cuda()
        pts, lab = feed.batch(rows)
        assert pts.is_cuda and pts.shape == (len(rows), 128, 3) and lab.dtype == torch.int64
        assert np.array_equal(pts.cpu().numpy(), g[f"mn.{part}.points"]) and np.array_equal(lab.cpu().numpy(), g[f"mn.{part}.labels"])
    feed = D.ResidentModelNet40(train=True, num_points=128, batch_size=4, randomize_data=True, use_normals=True, root=str(tmp_path))
    np.random.seed(77)
    orders = []
    for _ in range(3):                                    # ModelNet40Data.randomize's shuffles, replayed
        o = np.arange(128)
        np.random.shuffle(o)
        orders.append(o)
    rows = torch.as_tensor(_ref_rows(g, "train")[[0, 3, 8]]).cuda()
    pts, _ = feed.batch(rows, torch.as_tensor(np.stack(orders)).cuda())
    assert np.array_equal(pts.cpu().numpy(), g["mn.rand.points"])
    seen = 0
    for pts, lab in feed:                                 # an epoch: every cloud at most once, shuffled rows of the right clouds
        assert pts.shape == (4, 128, 6) and lab.shape == (4, 1)
        seen += 4
    assert seen == (9 // 4) * 4
    # scene flow from HBM: one launch per batch, the reference's items bit for bit (np.mean's order replayed on the device)
    for part in ("train", "test"):
        feed = D.ResidentSceneflow(npoints=256, root=sf, partition=part, batch_size=2)
        order = [str(s) for s in g[f"sf.{part}.order"]]
        mine = [os.path.basename(p)[:-4] for p in D.list_sceneflow_files(sf, part)]
        scene = torch.as_tensor([mine.index(st) for st in order]).cuda()
        s1 = s2 = None
        if part == "train":
            np.random.seed(91)
            draws = [(np.random.choice(700, 256, replace=False), np.random.choice(700, 256, replace=False)) for _ in order]
            s1 = torch.as_tensor(np.stack([d[0] for d in draws])).cuda()
            s2 = torch.as_tensor(np.stack([d[1] for d in draws])).cuda()
        out = feed.batch(scene, s1, s2)
        for i in range(len(order)):
            for name, v in zip(("pos1", "pos2", "color1", "color2", "flow", "mask1"), out):
                want = g[f"sf.{part}.{i}.{name}"]
                got = v[i].cpu().numpy()
                assert got.dtype == want.dtype and np.array_equal(got, want), (part, i, name)
        n = 0
        for item in feed:
            assert item[0].shape == (2, 256, 3) and item[5].dtype == torch.bool
            assert float(item[0].mean(dim=1).abs().max()) < 1e-4          # centred on the sampled first cloud
            n += 1
        assert n == len(order) // 2
