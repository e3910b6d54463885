#!/usr/bin/env python3
"""Wall time vs GPU time of training steps on the HIP training path: DGCNN features + max-pool loss,
PCN + Chamfer (examples/train_pcn.py:70-91).  Prints wall per step, the sum of kernel times per step and the top kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name, make, step, iters=5):
    from torch.profiler import ProfilerActivity, profile
    net, opt, data = make()
    for _ in range(3):
        step(net, opt, data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step(net, opt, data)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            step(net, opt, data)
        torch.cuda.synchronize()
    ka = prof.key_averages()
    gpu = sum(k.self_device_time_total for k in ka) / iters / 1e3
    n = sum(k.count for k in ka) / iters
    print(f"== {name}: wall {wall:.2f} ms per step, GPU kernels {gpu:.2f} ms per step in {n:.0f} launches")
    for k in sorted(ka, key=lambda k: -k.self_device_time_total)[:int(os.environ.get('TOPK', '12'))]:
        print(f"   {k.self_device_time_total / iters / 1e3:8.3f} ms  x{k.count / iters:5.1f}  {k.key[:100]}")


def main():
    which = sys.argv[1:] or ["dgcnn", "pcn", "pointnet"]
    from learning3d_amd.losses import ChamferDistanceLoss
    from learning3d_amd.models import DCP, DGCNN, PCN, FlowNet3D, PointNet
    torch.manual_seed(0)
    if "dgcnn" in which:
        def make():
            net = DGCNN(emb_dims=1024).cuda().train()
            return net, torch.optim.SGD(net.parameters(), lr=1e-3), torch.rand(32, 1024, 3, device="cuda")

        def step(net, opt, x):
            opt.zero_grad(set_to_none=True)
            loss = net(x).max(dim=2)[0].square().mean()
            loss.backward()
            opt.step()
        run("DGCNN train step (B 32, N 1024, emb 1024, train-mode BN)", make, step)
    if "pointnet" in which:
        def make():
            net = PointNet(emb_dims=1024).cuda().train()
            return net, torch.optim.SGD(net.parameters(), lr=1e-3), torch.rand(32, 1024, 3, device="cuda")

        def step(net, opt, x):
            opt.zero_grad(set_to_none=True)
            loss = net(x).max(dim=2)[0].square().mean()
            loss.backward()
            opt.step()
        run("PointNet train step (B 32, N 1024)", make, step)
    if "pcn" in which:
        def make():
            net = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).cuda().train()
            x = torch.rand(32, 2048, 3, device="cuda") - 0.5
            return net, torch.optim.Adam(net.parameters(), lr=1e-4), (x, torch.rand(32, 16384, 3, device="cuda") - 0.5)

        cd = ChamferDistanceLoss()

        def step(net, opt, data):
            x, gt = data
            opt.zero_grad(set_to_none=True)
            out = net(x)
            loss = cd(out["coarse_output"], gt[:, :1024]) + cd(out["fine_output"], gt)
            loss.backward()
            opt.step()
        run("PCN train step (B 32, 2048 -> 1024 / 16384, Chamfer on both outputs)", make, step)


def extra(which):
    from learning3d_amd.models import DCP, DGCNN, FlowNet3D
    if "flownet" in which:
        def make():
            net = FlowNet3D().cuda().train()
            g = torch.Generator().manual_seed(0)
            p1 = torch.rand(8, 3, 8192, generator=g).cuda(); p2 = torch.rand(8, 3, 8192, generator=g).cuda()
            return net, torch.optim.Adam(net.parameters(), lr=1e-3), (p1, p2, torch.rand(8, 3, 8192, generator=g).cuda())

        def step(net, opt, data):
            p1, p2, flow = data
            opt.zero_grad(set_to_none=True)
            loss = (net(p1, p2, p1, p2) - flow).square().mean()
            loss.backward()
            opt.step()
        run("FlowNet3D train step (B 8 per GPU, N 8192, train-mode BN; examples/train_flownet.py)", make, step, iters=3)
    if "dcp" in which:
        def make():
            net = DCP(feature_model=DGCNN(emb_dims=512), cycle=False).cuda().train()
            g = torch.Generator().manual_seed(0)
            t = (torch.rand(int(os.environ.get("DCP_B", "8")), 1024, 3, generator=g) - 0.5).cuda()
            return net, torch.optim.Adam(net.parameters(), lr=1e-3), (t, (t + 0.05).contiguous())

        def step(net, opt, data):
            t, s_ = data
            opt.zero_grad(set_to_none=True)
            out = net(t, s_)
            loss = (out["est_R"] - torch.eye(3, device="cuda")).square().mean() + out["est_t"].square().mean()
            loss.backward()
            opt.step()
        run(f"DCP-v2 train step (B {os.environ.get('DCP_B', '8')}, N 1024, emb 512; examples/train_dcp.py)", make, step, iters=3)


if __name__ == "__main__":
    if "--only" not in sys.argv:                   # `--only dcp flownet`: just the named extra steps
        main()
    extra([a for a in sys.argv[1:] if a != "--only"])
