"""Does the ORDER of the step's independent kernels matter?  The DGCNN chain (kNN -> EdgeConv -> conv5) and the Chamfer pair
(search -> loss tail) share nothing; captured as hipGraphs in several orders and replayed, time per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learning3d_amd.utils as U
from learning3d_amd.models import DGCNN, _fused
from learning3d_amd.losses.chamfer_distance import ChamferDistance, chamfer_loss_local
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda(); a = torch.rand((32, 1024, 3), generator=g).cuda(); b = torch.rand((32, 1024, 3), generator=g).cuda()
net = DGCNN(emb_dims=1024).cuda().eval(); cd = ChamferDistance()
xt = x.permute(0, 2, 1)
packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
w5, s5, b5, w5s, w5f = net._conv5_folded()
knn = lambda st: st.__setitem__("idx", U.knn(xt, 20))
ec = lambda st: st.__setitem__("img", _fused.edgeconv_forward(x, st["idx"], packed, planes=True, v2=True))
c5 = lambda st: st.__setitem__("feat", _fused.pointwise_conv_f16(st["img"], 32, 1024, w5f, 512, 1024, s5, b5, relu=True))
ch = lambda st: st.__setitem__("d", cd(a, b))
ls = lambda st: st.__setitem__("part", chamfer_loss_local(*st["d"]))
orders = {"knn ec c5 | ch ls (bench)": [knn, ec, c5, ch, ls], "ch ls | knn ec c5": [ch, ls, knn, ec, c5],
          "knn | ch ls | ec c5": [knn, ch, ls, ec, c5], "knn ch | ec | ls c5": [knn, ch, ec, ls, c5],
          "ch knn | ec c5 | ls": [ch, knn, ec, c5, ls], "knn ec | ch ls | c5": [knn, ec, ch, ls, c5]}
with torch.no_grad():
    graphs = {}
    for name, seq in orders.items():
        st = {}
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                for f in seq: f(st)
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for f in seq: f(st)
        graphs[name] = (gr, st)
    for rnd in range(3):
        for name, (gr, st) in graphs.items():
            for _ in range(100): gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(400): gr.replay()
            e1.record(); torch.cuda.synchronize()
            print(f"round {rnd}  {name:28s} {e0.elapsed_time(e1) / 400 * 1e3:8.1f} us/step")
