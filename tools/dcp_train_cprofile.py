#!/usr/bin/env python3
"""Where the HOST time of a DCP-v2 training step goes (cProfile over 5 steps after warm-up) and the step's wall time per
L3D_TRAIN_LINEAR route.  Diagnostic, not a product path."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning3d_amd.models import DCP, DGCNN  # noqa: E402

torch.manual_seed(0)
net = DCP(feature_model=DGCNN(emb_dims=512), cycle=False).cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(0)
t = (torch.rand(8, 1024, 3, generator=g) - 0.5).cuda()
s_ = (t + 0.05).contiguous()


def step():
    opt.zero_grad(set_to_none=True)
    out = net(t, s_)
    loss = (out["est_R"] - torch.eye(3, device="cuda")).square().mean() + out["est_t"].square().mean()
    loss.backward()
    opt.step()


for _ in range(12):                      # the caching allocator settles over the first ~10 steps (134 MB attention maps kept for the backward)
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
host = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10 * 1e3
print(f"route {os.environ.get('L3D_TRAIN_LINEAR', 'rows')}: host issue time {host:.2f} ms per step, wall {wall:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 22)
