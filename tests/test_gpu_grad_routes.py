"""Which route a reference user gets, and that gradients through it are right.

The reference's scripts run `model.eval()` with grad mode ON and never use torch.no_grad() (examples/test_pointnet.py:31-60,
examples/test_dcp.py:43-73, examples/test_pcn.py); examples/train_pcn.py:70-91 trains a network without BatchNorm.  These
tests drive the models exactly like that and assert (1) through the C-ABI launch log that the fused matrix-core kernels
served the forward, (2) outputs equal the no_grad results and the reference goldens, (3) `.backward()` gradients against an
fp64 evaluation of the reference's op sequence, (4) an fp16-range overflow is repaired inside the same call.
Tolerances are written where they are applied."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * (hi - lo) + lo).numpy()


class launch_log:
    def __enter__(self):
        from learning3d_amd import _lib
        self.lib = _lib
        _lib.LAUNCH_LOG = []
        return _lib.LAUNCH_LOG

    def __exit__(self, *exc):
        self.lib.LAUNCH_LOG = None
        return False


def _load(net, g):
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")}
    net.load_state_dict(sd)
    return net.cuda().eval()


def _rel(a, b):
    """max |a - b| over the scale max |b| (both numpy, b the fp64 truth)"""
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# --------------------------------------------------------------------------------------------- (1) + (2): the route
def test_eval_with_grad_enabled_runs_the_fused_kernels_dgcnn(golden):
    """examples/test_pointnet.py's call pattern on DGCNN: eval(), grad mode on, no no_grad."""
    from learning3d_amd.models import DGCNN
    g = golden("dgcnn_emb64")
    net = _load(DGCNN(emb_dims=64), g)
    assert torch.is_grad_enabled() and all(p.requires_grad for p in net.parameters())
    with launch_log() as log:
        out = net(dev(g["x"]))
    assert "l3d_knn_graph" in log and any(n.startswith("l3d_edgeconv_forward") for n in log), log
    assert not any(n in log for n in ("l3d_bn_act_forward", "l3d_graph_feature")), log      # the per-layer route did not run
    assert out.requires_grad                                                              # and the result is differentiable
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        assert torch.equal(net(dev(g["x"])), out.detach())
    # benchmark shape: the f16x2 EdgeConv kernel and the f16x2 conv5 are what runs
    torch.manual_seed(1)
    big = DGCNN(emb_dims=1024).cuda().eval()
    x = dev(rand((4, 1024, 3), 0))
    with launch_log() as log:
        y = big(x)
    assert log.count("l3d_edgeconv_forward_f16b") == 1 and log.count("l3d_pointwise_conv_f16[two-plane]") == 1, log    # the two-plane kernels
    with torch.no_grad():
        assert torch.equal(big(x), y.detach())


def test_eval_with_grad_enabled_runs_the_fused_kernels_other_models(golden):
    from learning3d_amd.models import DCP, DGCNN, PCN, Classifier, PointNet, FlowNet3D
    # PointNet classifier (config 1's pattern, examples/test_pointnet.py:98-118)
    torch.manual_seed(2)
    clf = Classifier(feature_model=PointNet(emb_dims=1024, use_bn=True)).cuda().eval()
    x = dev(rand((4, 1024, 3), 3, -1, 1))
    with launch_log() as log:
        logits = clf(x)
    assert any(n.startswith("l3d_pointwise_conv") for n in log) and "l3d_bn_act_forward" not in log, log
    with torch.no_grad():
        assert torch.equal(clf(x), logits.detach())
    # PCN, in train() mode as examples/train_pcn.py:70-91 runs it (no BatchNorm, no dropout: the same pure function)
    torch.manual_seed(3)
    pcn = PCN(emb_dims=1024, num_coarse=64, grid_size=2, detailed_output=True).cuda().train()
    xp = dev(rand((2, 512, 3), 4, -0.5, 0.5))
    with launch_log() as log:
        out = pcn(xp)
    assert "l3d_first_layer_f16_planes" in log and "l3d_pointwise_conv_f16[pool]" in log and "l3d_fold_mlp_f16" in log, log
    assert "l3d_bn_act_forward" not in log
    assert out["fine_output"].requires_grad and pcn.coarse_output is out["coarse_output"]
    with torch.no_grad():
        ref = pcn(xp)
    for k in out:
        assert torch.equal(out[k].detach(), ref[k]), k
    # DCP (examples/test_dcp.py:43-73): fused attention + soft correspondences + Kabsch
    gd = golden("dcp_emb64")
    dcp = _load(DCP(DGCNN(emb_dims=64)), gd)
    with launch_log() as log:
        o = dcp(dev(gd["template"]), dev(gd["source"]))
    assert any(n.startswith("l3d_soft_correspondence") for n in log) and "l3d_kabsch" in log, log
    np.testing.assert_allclose(o["est_R"].detach().cpu().numpy(), gd["est_R"], atol=1e-5)
    np.testing.assert_allclose(o["est_t"].detach().cpu().numpy(), gd["est_t"], atol=1e-5)
    assert o["est_R"].requires_grad
    o["est_R"].sum().backward()                                          # the whole chain back-propagates
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in dcp.parameters())
    assert dcp.emb_nn.conv1.weight.grad is not None and float(dcp.emb_nn.conv1.weight.grad.abs().max()) > 0
    # DCP at emb 512 (d_k = 128: the flash attention kernel and the f16x2 Linear layers)
    torch.manual_seed(4)
    dcp512 = DCP(DGCNN(emb_dims=512)).cuda().eval()
    tmpl = dev(rand((2, 256, 3), 5, -0.5, 0.5))
    src = (tmpl @ dev(np.array([[0.8, -0.6, 0], [0.6, 0.8, 0], [0, 0, 1]], dtype=np.float32)).t() + 0.1).contiguous()
    with launch_log() as log:
        o5 = dcp512(tmpl, src)
    assert any(n.startswith("l3d_attention_forward") for n in log) and any(n.startswith("l3d_layernorm_planes") for n in log), sorted(set(log))
    with torch.no_grad():
        r5 = dcp512(tmpl, src)
    assert torch.equal(o5["est_R"].detach(), r5["est_R"]) and torch.equal(o5["r"].detach(), r5["r"])
    # FlowNet3D
    torch.manual_seed(0)
    fn = FlowNet3D().cuda().eval()
    gq = torch.Generator().manual_seed(3)
    pc1 = torch.clamp(torch.randn((2, 3, 2048), generator=gq), -2, 2).cuda()
    pc2 = (pc1 + 0.05 * torch.randn((2, 3, 2048), generator=gq).cuda()).contiguous()
    f1, f2 = torch.rand((2, 3, 2048), generator=gq).cuda(), torch.rand((2, 3, 2048), generator=gq).cuda()
    with launch_log() as log:
        sf = fn(pc1, pc2, f1, f2)
    assert "l3d_group_first_layer" in log and "l3d_bn_act_forward" not in log, log
    with torch.no_grad():
        assert torch.equal(fn(pc1, pc2, f1, f2), sf.detach())


# --------------------------------------------------------------------------------------------- (3): gradients
def test_max_over_last_matches_torch():
    """_train.max_over_last (l3d_max_last / l3d_max_last_backward: the max over the k neighbours of an EdgeConv layer in a training
    step) against torch's max: values, the first-maximum rule under exact ties (ReLU zeros, duplicated entries), gradients."""
    from learning3d_amd.models import _train
    g = torch.Generator().manual_seed(3)
    for shape in [(2, 64, 300, 20), (3, 5, 7, 3), (1, 8, 16, 256), (4, 33)]:
        x = torch.randn(shape, generator=g).cuda()
        x = torch.where(torch.rand(shape, generator=g).cuda() < 0.3, torch.zeros_like(x), x).relu()      # many exact ties at 0 and at equal maxima
        x[..., 1] = x[..., 0]
        a, b = x.clone().requires_grad_(), x.clone().requires_grad_()
        with _lib_log() as log:
            va = _train.max_over_last(a)
        assert log == ["l3d_max_last"], log
        vb, ib = b.max(dim=-1, keepdim=True)
        assert torch.equal(va, vb)
        w = torch.randn(va.shape, generator=g).cuda()
        (va * w).sum().backward(); (vb * w).sum().backward()
        assert torch.equal(a.grad, b.grad), shape
    nanrow = torch.tensor([[1.0, float("nan"), 3.0, 2.0]], device="cuda")
    assert torch.isnan(_train.max_over_last(nanrow)).all()


def test_layernorm_hip_forward_backward_vs_fp64():
    """The pointer network's LayerNorm with autograd live (utils/transformer.py:109-119: unbiased std, eps added to std): HIP forward
    + one-pass HIP backward (_train._LayerNormRef) against the reference's op sequence differentiated in fp64 -- y, dx, da, db within
    1e-5 of their scale -- and against the same sequence in fp32 torch ops; da / db bit-identical from run to run."""
    from learning3d_amd.utils.transformer import LayerNorm
    g = torch.Generator().manual_seed(5)
    for shape in [(2, 256, 512), (3, 77, 64), (1, 5, 2048), (4, 100, 12), (8, 1024, 512)]:
        C_ = shape[-1]
        ln = LayerNorm(C_).cuda()
        with torch.no_grad():
            ln.a_2.copy_(torch.randn(C_, generator=g) * 0.5 + 1.0)
            ln.b_2.copy_(torch.randn(C_, generator=g) * 0.3)
        x = (torch.randn(shape, generator=g) * 2.0 + 0.7).cuda().requires_grad_()
        w = torch.randn(shape, generator=g).cuda()
        with _lib_log() as log:
            y = ln(x)
            (y * w).sum().backward()
        assert log == ["l3d_layernorm_planes[values]", "l3d_layernorm_ref_backward"], log
        x64 = x.detach().double().requires_grad_()
        a64, b64 = ln.a_2.detach().double().requires_grad_(), ln.b_2.detach().double().requires_grad_()
        y64 = a64 * (x64 - x64.mean(-1, keepdim=True)) / (x64.std(-1, keepdim=True) + ln.eps) + b64
        (y64 * w.double()).sum().backward()
        for name, got, want in (("y", y.detach(), y64.detach()), ("dx", x.grad, x64.grad), ("da", ln.a_2.grad, a64.grad),
                                ("db", ln.b_2.grad, b64.grad)):
            err = float((got.double() - want).abs().max())
            assert err <= 1e-5 * float(want.abs().max()), (shape, name, err, float(want.abs().max()))
        da1, db1 = ln.a_2.grad.clone(), ln.b_2.grad.clone()
        ln.zero_grad(); x.grad = None
        (ln(x) * w).sum().backward()
        assert torch.equal(da1, ln.a_2.grad) and torch.equal(db1, ln.b_2.grad)
    # switched off: the reference's torch ops
    from learning3d_amd.models import _fused
    old = _fused.TRAIN_HIP
    _fused.TRAIN_HIP = False
    try:
        with _lib_log() as log:
            (ln(x) * w).sum().backward()
        assert log == [], log
    finally:
        _fused.TRAIN_HIP = old


@pytest.mark.parametrize("route", ["rows", "conv", "torch"])
def test_pointer_network_training_gradients_vs_fp64(route, monkeypatch):
    """utils/transformer.py Transformer with autograd live (examples/train_dcp.py): LayerNorm on its HIP forward / backward kernels;
    the nn.Linear layers and the attention core (q k^T, softmax, p v) on l3d_bmm_f32 / l3d_softmax_rows, forward and backward
    (TRAIN_LINEAR "rows", the default), or the Linear layers on the channel-first conv / dgrad / wgrad kernels ("conv"), or both on
    torch / rocBLAS ("torch").  Outputs and every parameter / input gradient against the same module in fp64 (its torch route, which
    is the reference's op sequence): within 2e-5 of the gradient's scale (gradients that are zero by symmetry -- the key
    projection's bias under the softmax -- against the largest gradient)."""
    import copy
    from learning3d_amd.utils import transformer as T
    monkeypatch.setattr(T, "TRAIN_LINEAR", route)
    torch.manual_seed(11)
    net = T.Transformer(64, 1, 0.0, 128, 4).cuda().train()
    net64 = copy.deepcopy(net).double()
    g = torch.Generator().manual_seed(12)
    src = torch.randn((2, 64, 256), generator=g).cuda().requires_grad_()
    tgt = torch.randn((2, 64, 256), generator=g).cuda().requires_grad_()
    w0, w1 = torch.randn((2, 64, 256), generator=g).cuda(), torch.randn((2, 64, 256), generator=g).cuda()
    with _lib_log() as log:
        a, b = net(src, tgt)
        ((a * w0).sum() + (b * w1).sum()).backward()
    assert "l3d_layernorm_ref_backward" in log and ("l3d_wgrad" in log) == (route == "conv"), log
    # a train() module runs its forward ONCE, under autograd: no fused forward in front of the backward's own (_fused.checkpointed)
    assert "l3d_attention_forward_f16b" not in log and "l3d_layernorm_planes_cf" not in log, sorted(set(log))
    assert ("l3d_bmm_f32" in log and "l3d_softmax_rows" in log) == (route == "rows"), log
    if route == "rows":
        # 12 Linear layers x 2 passes x (forward, dgrad, wgrad; the bias gradient is l3d_colsum_rows) + 3 attention cores x 2 passes x
        # (2 forward + 4 backward products)
        assert log.count("l3d_bmm_f32") >= 12 * 2 * 3 + 6 * 6, log.count("l3d_bmm_f32")
        assert log.count("l3d_colsum_rows") >= 12 * 2, log.count("l3d_colsum_rows")
        assert log.count("l3d_softmax_rows") >= 12, log.count("l3d_softmax_rows")      # 6 cores, forward and backward (+ a recompute)
    s64, t64 = src.detach().double().requires_grad_(), tgt.detach().double().requires_grad_()
    a64, b64 = net64(s64, t64)
    ((a64 * w0.double()).sum() + (b64 * w1.double()).sum()).backward()
    gmax = max(float(q.grad.abs().max()) for q in net64.parameters())
    for name, got, want in [("src_emb", a.detach(), a64.detach()), ("tgt_emb", b.detach(), b64.detach()), ("d src", src.grad, s64.grad),
                            ("d tgt", tgt.grad, t64.grad)] + [(n, p.grad, q.grad) for (n, p), (_, q) in
                                                              zip(net.named_parameters(), net64.named_parameters())]:
        assert got is not None and want is not None, name
        err = float((got.double() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()) + 1e-7 * gmax, (name, err, float(want.abs().max()), gmax)


def test_bmm_kernel_every_layout_vs_fp64():
    """l3d_bmm_f32 (bmm.hip): operands read through their strides -- plain, transposed, sliced, head-split and broadcast views --
    at tile-edge shapes, with bias / ReLU / accumulate and split K.  An exact fp32 fma chain per element: against fp64 within the
    fp32 dot-product bound (K eps |a|.|b|), and no worse than torch's own fp32 matmul by more than 4x."""
    from learning3d_amd.models import _rows
    g = torch.Generator().manual_seed(5)

    def rnd(*shape):
        return torch.randn(shape, generator=g).cuda()

    cases = []
    a, b = rnd(3, 200, 70), rnd(3, 70, 150)
    cases.append(("plain", a, b))
    cases.append(("a transposed", rnd(3, 70, 200).transpose(1, 2), b))
    cases.append(("b transposed", a, rnd(3, 150, 70).transpose(1, 2)))
    cases.append(("both transposed, 2-D", rnd(129, 257).t(), rnd(130, 129).t()))
    big = rnd(2, 300, 4 * 32)                                                    # [B, N, h d] -> [B, h, N, d] views (attention heads)
    q = big.view(2, 300, 4, 32).transpose(1, 2)
    k = rnd(2, 260, 4 * 32).view(2, 260, 4, 32).transpose(1, 2)
    cases.append(("heads q k^T", q, k.transpose(-1, -2)))
    cases.append(("sliced", rnd(2, 100, 96)[:, 3:77, 5:69], rnd(2, 64, 40)))
    cases.append(("thin N=1", rnd(2, 500, 333), rnd(2, 333, 1)))
    cases.append(("thin M=1", rnd(1, 1, 1000), rnd(1, 1000, 48)))
    cases.append(("K=3", rnd(4, 3, 77).transpose(1, 2), rnd(4, 3, 90)))
    for name, x, y in cases:
        got = _rows.bmm(x, y)
        want = torch.matmul(x.double(), y.double())
        ref = torch.matmul(x, y)
        bound = torch.matmul(x.double().abs(), y.double().abs()) * (x.shape[-1] * 2.0 ** -24) + 1e-30
        err, err_t = (got.double() - want).abs(), (ref.double() - want).abs()
        assert bool((err <= bound).all()), (name, float((err / bound).max()))
        assert float(err.max()) <= 4 * float(err_t.max()) + 1e-6 * float(want.abs().max()), (name, float(err.max()), float(err_t.max()))
    # epilogue flags
    x, w, bias = rnd(300, 96), rnd(80, 96), rnd(80)
    got = _rows.bmm(x, w.t(), bias=bias, relu=True)
    want = torch.relu(x.double() @ w.double().t() + bias.double())
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    got = _rows.bmm(w, x.t(), bias=bias, bias_axis="m", alpha=0.5)
    want = 0.5 * (w.double() @ x.double().t()) + bias.double()[:, None]
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    acc = rnd(300, 80)
    want = acc.double() + x.double() @ w.double().t()
    _rows.bmm(x, w.t(), out=acc, accumulate=True)
    assert float((acc.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # a strided destination (columns of a wider tensor)
    wide = torch.zeros((300, 200), device="cuda")
    _rows.bmm(x, w.t(), out=wide[:, 40:120])
    assert float((wide[:, 40:120].double() - x.double() @ w.double().t()).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float(wide[:, :40].abs().max()) == 0 and float(wide[:, 120:].abs().max()) == 0
    # split K (a weight gradient: K = rows): same value every run, and within the bound
    gy, xx = rnd(5000, 96), rnd(5000, 64)
    want = gy.double().t() @ xx.double()
    r1 = _rows.bmm(gy.t(), xx, parts=_rows._split_parts(96, 64, 5000))
    r2 = _rows.bmm(gy.t(), xx, parts=_rows._split_parts(96, 64, 5000))
    assert _rows._split_parts(96, 64, 5000) > 1 and torch.equal(r1, r2)
    assert float((r1.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_colsum_rows_vs_fp64_and_repeatable():
    """l3d_colsum_rows (the bias gradient of an nn.Linear over rows, db = 1^T g): against fp64 within the fp32 summation bound, the same
    bits on every run, rows read through their stride (a column slice of a wider tensor), sizes off the 128-row / 256-column tiles."""
    from learning3d_amd.models import _rows
    g = torch.Generator().manual_seed(9)
    for R, Cn in ((1, 1), (127, 5), (1000, 300), (32768, 512), (5001, 1024)):
        x = torch.randn((R, Cn + 7), generator=g).cuda()
        for v in (x[:, :Cn], x[:, 3:3 + Cn], x[:, :Cn].contiguous()):
            got = _rows.colsum(v)
            assert torch.equal(got, _rows.colsum(v))
            want = v.double().sum(0)
            bound = v.double().abs().sum(0) * (R * 2.0 ** -24) + 1e-30
            assert bool(((got.double() - want).abs() <= bound).all()), (R, Cn)
    lin = torch.nn.Linear(40, 24).cuda()
    xin = torch.randn((333, 40), generator=g).cuda()
    w = torch.randn((333, 24), generator=g).cuda()
    (_rows.linear(xin, lin, relu=True) * w).sum().backward()
    want = ((torch.relu(lin(xin)) > 0) * w).double().sum(0)
    assert float((lin.bias.grad.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_linear_rows_f16x2_route_vs_fp64():
    """models/_rows.linear at the sizes of a DCP training step (rows % 256 == 0, Cout % 256 == 0): forward and dgrad as f16x2 products with the
    batch's rows as the kernel's weight operand (l3d_split_f16_operand + l3d_pointwise_conv_f16 with TWO_PLANE | SHIFT_N), weight and bias
    gradients on l3d_bmm_f32 / l3d_colsum_rows.  Outputs and all gradients against fp64 at the fp32 dot-product level; the three
    projections of one input split it once; L3D_TRAIN_GEMM=fp32 (the module switch) takes l3d_bmm_f32 for everything."""
    from learning3d_amd.models import _rows
    g = torch.Generator().manual_seed(21)
    for (R, Cin, Cout, relu, scale) in ((4096, 512, 256, False, 1.0), (8192, 256, 512, True, 1e-3), (4096, 1024, 512, True, 30.0)):
        lin = torch.nn.Linear(Cin, Cout).cuda()
        x = (torch.randn((R, Cin), generator=g) * scale).cuda().requires_grad_()
        wgt = torch.randn((R, Cout), generator=g).cuda()
        with _lib_log() as log:
            y = _rows.linear(x, lin, relu=relu)
            (y * wgt).sum().backward()
        assert log.count("l3d_pointwise_conv_f16[rows]") == 2 and log.count("l3d_bmm_f32") == 1 and "l3d_colsum_rows" in log, log
        lin64 = torch.nn.Linear(Cin, Cout).double().cuda()
        lin64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
        x64 = x.detach().double().requires_grad_()
        y64 = lin64(x64)
        if relu:                                                   # on the fp32 run's own branches
            y64 = y64 * (y.detach() > 0)
        (y64 * wgt.double()).sum().backward()
        ref = (x.detach() @ lin.weight.detach().t() + lin.bias.detach())
        e_ref = float((ref.double() - lin64(x64).detach()).abs().max())
        e_got = float(((y.detach().double() - y64.detach()).abs() * (y.detach() > 0 if relu else 1)).max())
        assert e_got <= 2.0 * e_ref + 1e-6 * float(y64.abs().max()), (R, Cin, Cout, e_got, e_ref)
        for name, got, want in (("dx", x.grad, x64.grad), ("dW", lin.weight.grad, lin64.weight.grad), ("db", lin.bias.grad, lin64.bias.grad)):
            err = float((got.double() - want).abs().max())
            assert err <= 2e-5 * float(want.abs().max()), (name, R, Cin, Cout, err, float(want.abs().max()))
    # three layers on one input: one split of the rows
    lins = [torch.nn.Linear(512, 512).cuda() for _ in range(3)]
    x = torch.randn((4096, 512), generator=g).cuda()
    with _lib_log() as log, torch.enable_grad():
        outs = [_rows.linear(x.requires_grad_(), l) for l in lins]
    assert log.count("l3d_split_f16_operand") == 1 + 3, log          # the rows once, each weight matrix once
    prev, _rows.TRAIN_GEMM = _rows.TRAIN_GEMM, "fp32"
    try:
        with _lib_log() as log:
            y32 = _rows.linear(x, lins[0])
        assert "l3d_pointwise_conv_f16[rows]" not in log and "l3d_bmm_f32" in log
    finally:
        _rows.TRAIN_GEMM = prev
    assert float((y32 - outs[0]).abs().max()) <= 1e-5 * float(y32.abs().max())


def test_softmax_rows_forward_backward_vs_fp64():
    from learning3d_amd.models import _rows
    g = torch.Generator().manual_seed(6)
    for rows, cols, scale in ((37, 1024, 0.0884), (5, 3, 1.0), (3, 8192, 0.5), (64, 300, 2.0)):
        x = (torch.randn((rows, cols), generator=g) * 3).cuda().requires_grad_()
        w = torch.randn((rows, cols), generator=g).cuda()
        p = _rows.softmax_rows(x, scale)
        (p * w).sum().backward()
        x64 = x.detach().double().requires_grad_()
        p64 = torch.softmax(x64 * scale, dim=-1)
        (p64 * w.double()).sum().backward()
        assert float((p.double() - p64).abs().max()) <= 1e-6
        assert float((x.grad.double() - x64.grad).abs().max()) <= 1e-5 * float(x64.grad.abs().max()) + 1e-9


def test_square_distance_index_points_and_svd_head_gradients_on_hip():
    """utils/model_common_utils.square_distance / index_points and the SVD head's score route with autograd live: forward through
    the kernels, gradients through l3d_bmm_f32 / l3d_scatter_add_det; against torch autograd on the reference's op sequence in fp64."""
    from learning3d_amd.utils import model_common_utils as M
    from learning3d_amd.utils.svd import SVDHead
    g = torch.Generator().manual_seed(7)
    src = torch.randn((3, 200, 3), generator=g).cuda().requires_grad_()
    dst = torch.randn((3, 150, 3), generator=g).cuda().requires_grad_()
    w = torch.randn((3, 200, 150), generator=g).cuda()
    with _lib_log() as log:
        d = M.square_distance(src, dst)
        (d * w).sum().backward()
    assert "l3d_square_distance" in log and log.count("l3d_bmm_f32") == 4, log
    s64, d64 = src.detach().double().requires_grad_(), dst.detach().double().requires_grad_()
    ref = -2 * torch.matmul(s64, d64.permute(0, 2, 1)) + (s64 ** 2).sum(-1).view(3, 200, 1) + (d64 ** 2).sum(-1).view(3, 1, 150)
    (ref * w.double()).sum().backward()
    assert float((d.double() - ref).abs().max()) <= 1e-5
    for got, want in ((src.grad, s64.grad), (dst.grad, d64.grad)):
        assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # index_points: duplicates in idx (the scatter adds them up, in a fixed order)
    pts = torch.randn((2, 50, 7), generator=g).cuda().requires_grad_()
    idx = torch.randint(0, 50, (2, 30, 4), generator=g).cuda()
    wv = torch.randn((2, 30, 4, 7), generator=g).cuda()
    with _lib_log() as log:
        out = M.index_points(pts, idx)
        (out * wv).sum().backward()
    assert "l3d_index_points" in log and "l3d_scatter_add_det" in log, log
    p64 = pts.detach().double().requires_grad_()
    bi = torch.arange(2, device="cuda").view(2, 1, 1).expand_as(idx)
    ref = p64[bi, idx, :]
    (ref * wv.double()).sum().backward()
    assert torch.equal(out.detach(), ref.float()) and float((pts.grad.double() - p64.grad).abs().max()) <= 1e-6
    # the SVD head with gradients flowing into the embeddings (examples/train_dcp.py)
    head = SVDHead(emb_dims=64, input_shape="bnc").cuda()
    se = torch.randn((2, 64, 128), generator=g).cuda().requires_grad_()
    te = torch.randn((2, 64, 128), generator=g).cuda().requires_grad_()
    xs, xt = torch.randn((2, 128, 3), generator=g).cuda(), torch.randn((2, 128, 3), generator=g).cuda()
    with _lib_log() as log:
        R, t = head(se, te, xs, xt)
        ((R * torch.arange(9., device="cuda").view(1, 3, 3)).sum() + t.sum()).backward()
    assert "l3d_bmm_f32" in log and "l3d_softmax_rows" in log, log
    import math
    s2, t2 = se.detach().double().requires_grad_(), te.detach().double().requires_grad_()
    scores = torch.softmax(torch.matmul(s2.transpose(2, 1), t2) / math.sqrt(64), dim=2)
    corr = torch.matmul(xt.double().permute(0, 2, 1), scores.transpose(2, 1))
    sd = xs.double().permute(0, 2, 1)
    sc_, cc_ = sd - sd.mean(2, keepdim=True), corr - corr.mean(2, keepdim=True)
    H = sc_ @ cc_.transpose(2, 1)
    Rs = []
    for i in range(2):                                                       # utils/svd.py:38-49 of the reference
        u, _, v = torch.svd(H[i])
        r = v @ u.t()
        if float(torch.det(r.detach())) < 0:
            r = (v @ torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64, device="cuda"))) @ u.t()
        Rs.append(r)
    R2 = torch.stack(Rs)
    tt2 = (torch.matmul(-R2, sd.mean(2, keepdim=True)) + corr.mean(2, keepdim=True)).view(2, 3)
    ((R2 * torch.arange(9., device="cuda", dtype=torch.float64).view(1, 3, 3)).sum() + tt2.sum()).backward()
    assert float((R.double() - R2).abs().max()) <= 1e-5
    for got, want in ((se.grad, s2.grad), (te.grad, t2.grad)):
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-9


class _lib_log:
    def __enter__(self):
        from learning3d_amd import _lib
        _lib.LAUNCH_LOG = []
        return _lib.LAUNCH_LOG

    def __exit__(self, *exc):
        from learning3d_amd import _lib
        _lib.LAUNCH_LOG = None


def _dgcnn_layers_fp32(net, x):
    """The per-layer route of DGCNN._forward (models/dgcnn.py here; reference models/dgcnn.py:25-49) spelled out on the HIP
    layer kernels, keeping what the fp64 evaluation needs: the graph feature, every layer's ReLU mask and the arg-max of every
    max over k.  Runs on a deep copy (train-mode BatchNorm updates its running statistics)."""
    import copy
    from learning3d_amd.models import _train
    from learning3d_amd.utils import get_graph_feature
    net = copy.deepcopy(net)
    B, N, _ = x.shape
    with torch.no_grad():
        h = get_graph_feature(x.permute(0, 2, 1)).contiguous()
        feat, masks, args, outs = h, [], [], []
        for conv, bn in ((net.conv1, net.bn1), (net.conv2, net.bn2), (net.conv3, net.bn3), (net.conv4, net.bn4)):
            h = _train.conv_bn_act(h, conv, bn)
            masks.append(h > 0)
            v, a = h.max(dim=-1, keepdim=True)
            args.append(a); outs.append(v)
        out = _train.conv_bn_act(torch.cat(outs, dim=1), net.conv5, net.bn5)
        masks.append(out > 0)
    return feat, masks, args, out.view(B, -1, N)


def _dgcnn_fp64(net, x, patterns=None):
    """fp64 evaluation of models/dgcnn.py:25-49 on the HIP kNN graph.  With `patterns` (from _dgcnn_layers_fp32) every ReLU and
    max over k takes the branch the fp32 run took -- the function the fp32 backward differentiates; an element whose
    pre-activation is within fp32 rounding of zero may sit on the other side in fp64, and a gradient bar of 1e-5 cannot
    survive even one such flip (measured: single flips move bn.bias gradients by 1e-5 ... 1e-3 of their scale)."""
    from learning3d_amd.utils import get_graph_feature
    B, N, _ = x.shape
    n64 = type(net)(emb_dims=net.emb_dims).cuda().double()
    n64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()})
    n64.train(net.training)
    x64 = x.double().requires_grad_()
    with torch.no_grad():
        f32 = get_graph_feature(x.permute(0, 2, 1)).contiguous()
        idx = None
    # the graph feature as a differentiable function of x64: cat(neighbour, centre), model_common_utils.py:146-154
    from learning3d_amd.utils import knn
    with torch.no_grad():
        idx = knn(x.permute(0, 2, 1), 20)
    nb = torch.gather(x64.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, 20, 3))
    ce = x64.unsqueeze(2).expand(B, N, 20, 3)
    h = torch.cat([nb, ce], dim=3).permute(0, 3, 1, 2)
    assert torch.equal(h.detach().float(), f32)
    outs = []
    for i, (conv, bn) in enumerate(((n64.conv1, n64.bn1), (n64.conv2, n64.bn2), (n64.conv3, n64.bn3), (n64.conv4, n64.bn4))):
        z = bn(conv(h))
        if patterns is None:
            h = F.relu(z)
            outs.append(h.max(dim=-1, keepdim=True)[0])
        else:
            h = z * patterns[1][i].double()
            outs.append(torch.gather(h, 3, patterns[2][i]))
    z = n64.bn5(n64.conv5(torch.cat(outs, dim=1)))
    out = (F.relu(z) if patterns is None else z * patterns[1][4].double()).view(B, -1, N)
    return n64, x64, out


def _grad_errors(net, n64, x=None, x64=None):
    errs = {k: _rel(p.grad.double().cpu().numpy(), q.grad.cpu().numpy())
            for (k, p), (_, q) in zip(net.named_parameters(), n64.named_parameters())}
    if x is not None:
        errs["x"] = _rel(x.grad.double().cpu().numpy(), x64.grad.cpu().numpy())
    return errs


def test_eval_backward_matches_fp64_dgcnn():
    """eval-mode DGCNN, grad on: forward = fused kernels, backward = recomputation on the HIP conv / dgrad / wgrad kernels.
    Every parameter gradient and the input gradient against fp64 on the same ReLU / max branches: <= 1e-5 of the gradient's
    scale (max |truth|)."""
    from learning3d_amd.models import DGCNN
    torch.manual_seed(21)
    net = DGCNN(emb_dims=256).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
    x = dev(rand((4, 256, 3), 61)).requires_grad_()
    w = dev(np.random.default_rng(5).standard_normal((4, 256, 256)).astype(np.float32))
    with launch_log() as log:
        out = net(x)
        assert any(n.startswith("l3d_edgeconv_forward") for n in log)
        (out * w).sum().backward()
        assert "l3d_wgrad" in log and "l3d_bn_act_backward" in log, log       # the backward ran on the HIP layer kernels
    pat = _dgcnn_layers_fp32(net, x.detach())
    np.testing.assert_allclose(out.detach().cpu().numpy(), pat[3].cpu().numpy(), rtol=1e-4, atol=1e-5)   # fused vs per-layer
    n64, x64, o64 = _dgcnn_fp64(net, x.detach(), pat)
    np.testing.assert_allclose(out.detach().cpu().numpy(), o64.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    (o64 * w.double()).sum().backward()
    errs = _grad_errors(net, n64, x, x64)
    print("eval-backward relative errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= 1e-5, errs


def test_dgcnn_training_step_vs_fp64():
    """DGCNN in .train() (batch statistics): one forward + backward through the HIP layer kernels (_train.py) against the same
    step in fp64 on the same ReLU / max branches: loss, every parameter gradient, running statistics.  BatchNorm's backward
    cancels (g - mean g - zhat mean(g zhat)) and the weight gradient then sums dz x over all points: the backward kernels carry
    the per-channel constants in fp64 and the weight gradient adds its split-K pieces in fp64.  Bar: the tier's 1e-5 of each
    gradient's scale.  (Round 2 compared against the plain fp64 ReLU network and read single ReLU flips at rounding-level
    pre-activations as a 7e-4 error of the kernels; torch's own fp32 route shows the same flips, tools/grad_diag.py.)"""
    from learning3d_amd.models import DGCNN
    torch.manual_seed(13)
    net = DGCNN(emb_dims=256).cuda().train()
    x = dev(rand((4, 256, 3), 60))
    pat = _dgcnn_layers_fp32(net, x)
    n64, _, o64 = _dgcnn_fp64(net, x, pat)
    with launch_log() as log:
        out = net(x)
        loss = (out ** 2).mean()
        loss.backward()
    assert "l3d_channel_stats" in log and "l3d_wgrad" in log and "l3d_bn_backward_stats" in log, sorted(set(log))
    assert torch.equal(out.detach(), pat[3])                                  # the model IS that layer sequence
    loss64 = (o64 ** 2).mean()
    loss64.backward()
    assert abs(float(loss.detach()) - float(loss64.detach())) <= 1e-5 * max(1.0, abs(float(loss64.detach())))
    errs = _grad_errors(net, n64)
    print("DGCNN training-step relative gradient errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= 1e-5, errs
    for (k, a), (_, b) in zip(net.named_buffers(), n64.named_buffers()):
        if "running" in k:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_wgrad_kernel_vs_fp64_and_deterministic():
    """l3d_wgrad (split-K fp32 MFMA, pieces added in fp64) against an fp64 einsum, ragged shapes included; two runs give the
    same bits.  Bar: 2e-6 of max |dW| (fp32 products, <= 2048-term fp32 partial sums, one final rounding)."""
    from learning3d_amd.models import _train
    rng = np.random.default_rng(9)
    for (B, Cout, Cin, P) in [(4, 64, 6, 5120), (3, 128, 64, 1000), (2, 256, 128, 20480), (5, 100, 37, 333), (1, 1024, 512, 64),
                              (2, 3, 512, 4096)]:
        dz = dev(rng.standard_normal((B, Cout, P)).astype(np.float32))
        x = dev((rng.standard_normal((B, Cin, P)) + 0.5).astype(np.float32))
        got = _train.wgrad(dz, x)
        want = torch.einsum("bop,bip->oi", dz.double(), x.double())
        assert _rel(got.double().cpu().numpy(), want.cpu().numpy()) <= 2e-6, (B, Cout, Cin, P)
        assert torch.equal(got, _train.wgrad(dz, x))
        assert torch.equal(_train.wgrad(dz, x, pc=256), _train.wgrad(dz, x, pc=256))


def test_conv_layer_eval_bias_and_leaky_variants_vs_fp64():
    """_train.conv_bn_act in its three statistic modes (batch, running, none), with and without bias, ReLU / LeakyReLU / no
    activation, and linear_act: outputs and all gradients against fp64 torch.  Bar 1e-5 of each gradient's scale."""
    from learning3d_amd.models import _train
    from learning3d_amd.models.prnet import ACT_LRELU
    torch.manual_seed(31)
    cases = [("batch", True, 1), ("batch", False, ACT_LRELU), ("running", True, 1), ("running", False, 0), ("none", True, 1),
             ("none", True, 0), ("none", False, ACT_LRELU)]
    for (B, Cin, Cout, P) in [(3, 6, 64, 2560), (2, 128, 256, 512), (2, 259, 64, 300)]:
        for mode, bias, act in cases:
            conv = torch.nn.Conv1d(Cin, Cout, 1, bias=bias).cuda()
            bn = None
            if mode != "none":
                bn = torch.nn.BatchNorm1d(Cout).cuda()
                bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.3, 0.3)
                bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2.0)
                bn.train(mode == "batch")
            c64 = torch.nn.Conv1d(Cin, Cout, 1, bias=bias).cuda().double()
            c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
            b64 = None
            if bn is not None:
                b64 = torch.nn.BatchNorm1d(Cout).cuda().double()
                b64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
                b64.train(mode == "batch")
            x = (torch.randn(B, Cin, P, device="cuda") * 0.7 + 0.2)
            go = torch.randn(B, Cout, P, device="cuda")
            xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
            ya = _train.conv_bn_act(xa, conv, bn, relu=act, sync=False)
            z = c64(xb)
            z = b64(z) if b64 is not None else z
            yb = z if act == 0 else (F.relu(z) if act == 1 else F.leaky_relu(z, 0.2))
            np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
            ya.backward(go); yb.backward(go.double())
            pairs = [("x", xa.grad, xb.grad), ("W", conv.weight.grad, c64.weight.grad)]
            if bias:
                pairs.append(("bias", conv.bias.grad, c64.bias.grad))
            if bn is not None:
                pairs += [("gamma", bn.weight.grad, b64.weight.grad), ("beta", bn.bias.grad, b64.bias.grad)]
            for name, a, b in pairs:
                if mode == "batch" and name == "bias":
                    assert float(a.abs().max()) <= 1e-4 * float(go.abs().sum())     # cancels in front of batch statistics
                    continue
                assert _rel(a.double().cpu().numpy(), b.cpu().numpy()) <= 1e-5, ((B, Cin, Cout, P), mode, bias, act, name,
                                                                                 _rel(a.double().cpu().numpy(), b.cpu().numpy()))
            if mode == "batch":
                np.testing.assert_allclose(bn.running_mean.cpu().numpy(), b64.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(bn.running_var.cpu().numpy(), b64.running_var.cpu().numpy(), rtol=1e-5, atol=1e-6)
    lin = torch.nn.Linear(1024, 192).cuda()
    l64 = torch.nn.Linear(1024, 192).cuda().double()
    l64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    x = torch.randn(7, 1024, device="cuda")
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    ya, yb = _train.linear_act(xa, lin, relu=True), F.relu(l64(xb))
    np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    go = torch.randn(7, 192, device="cuda")
    ya.backward(go); yb.backward(go.double())
    for a, b in ((xa.grad, xb.grad), (lin.weight.grad, l64.weight.grad), (lin.bias.grad, l64.bias.grad)):
        assert _rel(a.double().cpu().numpy(), b.cpu().numpy()) <= 1e-5


def test_pcn_training_step_vs_fp64():
    """examples/train_pcn.py:70-91's step on the HIP path end to end: PCN.train() forward (fused f16x2 kernels) -> Chamfer loss
    (HIP) -> backward (recomputation on the HIP conv / dgrad / wgrad kernels + the HIP Chamfer backward), against the same
    step in fp64 torch (reference op order, models/pcn.py:110-153).  Bar: loss 1e-5 relative, every parameter gradient within
    1e-5 of its scale."""
    from learning3d_amd.losses import ChamferDistanceLoss
    from learning3d_amd.models import PCN
    torch.manual_seed(41)
    net = PCN(emb_dims=1024, num_coarse=64, grid_size=2, detailed_output=True).cuda().train()
    x = dev(rand((3, 512, 3), 42, -0.5, 0.5))
    gt = dev(rand((3, 1024, 3), 43, -0.5, 0.5))
    with launch_log() as log:
        out = net(x)
        loss = ChamferDistanceLoss()(gt, out["fine_output"]) + ChamferDistanceLoss()(gt, out["coarse_output"])
        loss.backward()
    assert "l3d_fold_mlp_f16" in log and "l3d_wgrad" in log and "l3d_chamfer_backward" in log, sorted(set(log))
    n64 = PCN(emb_dims=1024, num_coarse=64, grid_size=2, detailed_output=True).cuda().double()
    n64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    x64 = x.double().permute(0, 2, 1)
    h = n64.conv2(F.relu(n64.conv1(x64)))
    h = torch.cat([h, h.max(dim=2, keepdim=True)[0].expand(-1, -1, h.shape[2])], dim=1)
    gfeat = n64.conv4(F.relu(n64.conv3(h))).max(dim=2)[0]
    coarse = n64.linear3(F.relu(n64.linear2(F.relu(n64.linear1(gfeat))))).view(3, 64, 3)
    n64.num_points = 512
    fine = n64._fine_torch(coarse, gfeat)

    def cd64(a, b):
        d = torch.cdist(a, b) ** 2
        return 0.5 * (d.min(dim=2)[0].sqrt().mean() + d.min(dim=1)[0].sqrt().mean())
    loss64 = cd64(gt.double(), fine) + cd64(gt.double(), coarse)
    loss64.backward()
    assert abs(float(loss) - float(loss64)) <= 1e-5 * abs(float(loss64))
    errs = {k: _rel(p.grad.double().cpu().numpy(), q.grad.cpu().numpy())
            for (k, p), (_, q) in zip(net.named_parameters(), n64.named_parameters())}
    print("PCN training-step relative gradient errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= 1e-5, errs


def test_pointnet_eval_and_train_backward_vs_fp64():
    """PointNet (models/pointnet.py:51-73), with and without BatchNorm, eval and train: gradients through the HIP layers against
    fp64 on the same ReLU branches.  Bar 1e-5 of each gradient's scale (a conv bias in front of batch statistics has a zero
    gradient: exactly 0 here, rounding noise in torch)."""
    import copy
    from learning3d_amd.models import PointNet, _fused, _train
    for use_bn, training in ((False, False), (True, False), (True, True)):
        torch.manual_seed(51)
        net = PointNet(emb_dims=256, use_bn=use_bn).cuda()
        if use_bn:
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
        net.train(training)
        n64 = PointNet(emb_dims=256, use_bn=use_bn).double().cuda()
        n64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()})
        n64.train(training)
        x = dev(rand((4, 512, 3), 52, -1, 1))
        w = dev(np.random.default_rng(6).standard_normal((4, 256, 512)).astype(np.float32))
        masks, h, c = [], x.permute(0, 2, 1).contiguous(), copy.deepcopy(net)
        with torch.no_grad():
            for conv, bn in c._stack():
                h = _train.conv_bn_act(h, conv, bn)
                masks.append(h > 0)
        out = net(x)
        np.testing.assert_allclose(out.detach().cpu().numpy(), h.cpu().numpy(), rtol=1e-4, atol=1e-5)
        (out * w).sum().backward()
        h = x.double().permute(0, 2, 1)
        for (conv, bn), m in zip(n64._stack(), masks):
            h = conv(h)
            h = (bn(h) if bn is not None else h) * m.double()
        (h * w.double()).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), h.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
        errs = _grad_errors(net, n64)
        if use_bn and training:
            for k in [k for k in errs if k.startswith("conv") and k.endswith(".bias")]:
                assert float(dict(net.named_parameters())[k].grad.abs().max()) == 0.0
                del errs[k]
        assert max(errs.values()) <= 1e-5, (use_bn, training, errs)


# --------------------------------------------------------------------------------------------- (4): fp16 range
def test_f16_overflow_is_repaired_inside_the_same_call():
    """Unnormalised clouds (coordinates ~1e4) under eval BatchNorm overflow the f16x2 EdgeConv kernel's static exponents.  The
    first call already returns the right features (re-run on bf16x3 inside the call), like the reference for any input
    scale; "raise" turns the same event into L3DRangeError; "async" leaves it to check_range."""
    from learning3d_amd.models import DGCNN, _fused
    torch.manual_seed(71)
    net = DGCNN(emb_dims=256).cuda().eval()
    x = dev(rand((2, 256, 3), 72)) * 3.0e4
    assert _fused.gemm_arith() == "f16x2" and _fused.RANGE_POLICY == "retry"
    before = _fused.RANGE_RETRIES
    with torch.no_grad():
        out = net(x)
        torch.cuda.synchronize()
        assert _fused.RANGE_RETRIES == before + 1, "the overflow was not detected in the call that caused it"
        with _fused.arith("bf16x3"):
            want = net(x)
        assert torch.equal(out, want)
        with torch.enable_grad():
            _, _, o64 = _dgcnn_fp64(net, x)
        np.testing.assert_allclose(out.cpu().numpy(), o64.detach().cpu().numpy(), rtol=1e-4, atol=1e-5 * float(o64.abs().max()))
        _fused.check_range(sync=True)                                   # nothing left behind
        small = net(dev(rand((2, 256, 3), 73)))                         # in-range input: no retry
        assert _fused.RANGE_RETRIES == before + 1 and torch.isfinite(small).all()
        _fused.RANGE_POLICY = "raise"
        try:
            with pytest.raises(_fused.L3DRangeError):
                net(x)
        finally:
            _fused.RANGE_POLICY = "retry"
        _fused.check_range(sync=True)
    # grad mode on: the same repair inside the checkpointed forward
    out2 = net(x)
    assert torch.equal(out2.detach(), out) and _fused.RANGE_RETRIES == before + 2


def test_transforms_take_the_reference_datasets_cpu_inputs_and_hooks_see_layernorm_values():
    """ The reference's RegistrationData hands its transforms CPU tensors [N,3] (data_utils/dataloaders.py:290-296):
    accepted, computed on the device, returned on the CPU; PCRNetTransform keeps a fixed pose per sample index.  A forward hook
    on a sublayer's LayerNorm (or a subclassed feed-forward) must see real values, not the deferred-output buffer."""
    from learning3d_amd.ops.transform_functions import DCPTransform, PCRNetTransform, PNLKTransform
    from learning3d_amd.utils.transformer import Transformer
    t = torch.rand((64, 3)) - 0.5
    for tf in (DCPTransform(45, 1), PNLKTransform(0.8, True), PCRNetTransform(10, 45, 1)):
        src = tf(t)
        assert not src.is_cuda and src.shape == (64, 3) and not tf.igt.is_cuda and torch.isfinite(src).all()
    pcr = PCRNetTransform(5, 45, 1)
    pcr.index = 3
    a = pcr(t)
    pcr.index = 1
    pcr(t)
    pcr.index = 3
    assert torch.equal(pcr(t), a)                                       # the same pose for the same index
    torch.manual_seed(5)
    net = Transformer(512, 1, 0.0, 1024, 4).cuda().eval()
    x = torch.randn(2, 512, 256, device="cuda")
    with torch.no_grad():
        want = net(x, x)
        seen = []
        h = net.model.encoder.layers[0].sublayer[1].norm.register_forward_hook(lambda m, i, o: seen.append(o.clone()))
        got = net(x, x)
        h.remove()
    # the hooked module runs module by module, the plain one as a channel-first pass (Transformer._pass_cf): same values to fp32 rounding
    for g_, w_ in zip(got, want):
        assert (g_ - w_).abs().max().item() <= 2e-5 * w_.abs().max().item()
    ln = net.model.encoder.layers[0].sublayer[1].norm
    assert len(seen) == 2 and all(torch.isfinite(o).all() for o in seen)
    assert all(abs(float(o.mean())) < 0.1 and 0.5 < float(o.std()) < 2.0 for o in seen)     # normalised values, not scratch memory

