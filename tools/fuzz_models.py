#!/usr/bin/env python3
"""Random-shape sweep of the model forwards: the fused route (what a call takes by default) against the per-layer HIP route and
against the reference's op sequence on torch ops (every model file keeps it as the route for _fused.TRAIN_HIP = False; the
geometry ops underneath are the HIP ones in all three, they have no CPU form; the pointer network is pure torch and is compared
with its fp64 evaluation on the CPU).  Catches route-selection mistakes at shapes the fixed tests do not visit: point counts that are not multiples of the
GEMM tiles, tiny batches, all three GEMM arithmetics.  `python tools/fuzz_models.py [seed] [rounds]` on a GPU box."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd.models import DGCNN, PCN, PointNet, _fused  # noqa: E402
from learning3d_amd.utils.transformer import Transformer  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
worst = {}


def rec(name, got, want, rtol, atol_rel):
    want = want.astype(np.float64)
    err = np.abs(got.astype(np.float64) - want).max()
    bar = rtol * np.abs(want).max() * 0 + atol_rel * max(1e-30, np.abs(want).max())
    worst[name] = max(worst.get(name, 0.0), err / max(1e-30, np.abs(want).max()))
    assert err <= bar, (name, err, bar)


def randomise_bn(net):
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
            m.weight.data.uniform_(0.7, 1.3); m.bias.data.uniform_(-0.2, 0.2)


for it in range(rounds):
    B = int(rng.integers(1, 4)); N = int(rng.choice([64, 100, 256, 300, 512, 1000, 1024, 1280]))
    arith = str(rng.choice(["f16x2", "f16x2", "bf16x3", "fp32"]))
    x = torch.from_numpy(rng.uniform(-1, 1, (B, N, 3)).astype(np.float32))
    with _fused.arith(arith) if arith != "fp32" else _fused.arith("f16x2"):
        prev_split = _fused.SPLIT_BF16
        _fused.SPLIT_BF16 = arith != "fp32"
        try:
            for name, make in (("DGCNN", lambda: DGCNN(emb_dims=int(rng.choice([64, 512, 1024])))),
                               ("PointNet", lambda: PointNet(emb_dims=int(rng.choice([256, 1024])))),
                               ("PCN", lambda: PCN(emb_dims=1024, num_coarse=int(rng.choice([64, 256])), grid_size=2, detailed_output=True))):
                net = make().eval()
                randomise_bn(net)
                net = net.cuda()
                with torch.no_grad():
                    got = net(x.cuda())
                with _fused.per_layer_route(), torch.enable_grad():
                    lay = net(x.cuda().requires_grad_())
                    _fused.TRAIN_HIP = False
                    try:
                        want = net(x.cuda().requires_grad_())
                    finally:
                        _fused.TRAIN_HIP = True
                outs = [("out", got, want, lay)] if torch.is_tensor(got) else [(k_, got[k_], want[k_], lay[k_]) for k_ in got]
                for k_, g_, w_, l_ in outs:
                    rec(f"{name} fused vs torch ops ({arith})", g_.cpu().numpy(), w_.detach().cpu().numpy(), 0, 2e-4)
                    rec(f"{name} per-layer HIP vs torch ops", l_.detach().cpu().numpy(), w_.detach().cpu().numpy(), 0, 2e-4)
                _fused.check_range(sync=True)
            if arith == "f16x2":
                C = int(rng.choice([256, 512])); Nt = int(rng.choice([128, 256, 320, 512])); Ns = int(rng.choice([128, 256, 512]))
                net = Transformer(C, 1, 0.0, 2 * C, 4).eval()
                ref = copy.deepcopy(net).double()
                a = torch.from_numpy(rng.standard_normal((B, C, Ns)).astype(np.float32)); b = torch.from_numpy(rng.standard_normal((B, C, Nt)).astype(np.float32))
                with torch.no_grad():
                    want = ref(a.double(), b.double()); got = net.cuda()(a.cuda(), b.cuda())
                for g_, w_ in zip(got, want):
                    rec("Transformer vs fp64", g_.cpu().numpy(), w_.numpy(), 0, 2e-4)
        finally:
            _fused.SPLIT_BF16 = prev_split
    print(f"round {it}: B {B} N {N} {arith} ok", flush=True)
for k_, v_ in worst.items():
    print(f"{k_:40s} worst |err| / max|want| = {v_:.2e}")
print("fuzz_models OK")
