"""Why is the EdgeConv kernel 8-11 % slower inside the replayed step than back to back?  Its duration (events
around the launch, NO host synchronisation inside the loop: the GPU runs the sequence back to back like the step) behind
different predecessors -- in particular behind a kNN launch whose freshly written indices it reads, against a kNN launch that
writes a buffer it does not read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learning3d_amd.utils as U
from learning3d_amd.models import DGCNN, _fused

g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda()
net = DGCNN(emb_dims=1024).cuda().eval()
with torch.no_grad():
    xt = x.permute(0, 2, 1)
    idx0 = U.knn(xt, 20)
    packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
    w5, s5, b5, w5s, w5f = net._conv5_folded()
    img = _fused.edgeconv_forward(x, idx0, packed, planes=True, v2=True, unscaled=True)

    def ec(idx):
        return _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True, unscaled=True)

    def conv5(im):
        return _fused.pointwise_conv_f16(im, 32, 1024, w5f, 512, 1024, s5, b5, relu=True, unscaled=True)

    def run(name, body, iters=200, warm=60):
        evs = []
        for it in range(warm + iters):
            e = body()
            if it >= warm:
                evs.append(e)
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        print(f"{name:62s} median {ts[len(ts) // 2]:7.1f} us   p10 {ts[len(ts) // 10]:7.1f}   p90 {ts[9 * len(ts) // 10]:7.1f}", flush=True)

    def timed_ec(idx):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = ec(idx); e1.record()
        return (e0, e1), out

    def b_alone():
        return timed_ec(idx0)[0]

    def b_knn_fresh():
        i = U.knn(xt, 20)
        return timed_ec(i)[0]

    def b_knn_other():
        U.knn(xt, 20)
        return timed_ec(idx0)[0]

    def b_conv5_then():
        conv5(img)
        return timed_ec(idx0)[0]

    def b_step_clean():
        U.knn(xt, 20)
        ev, out = timed_ec(idx0)
        conv5(out)
        return ev

    def b_step_real():
        i = U.knn(xt, 20)
        ev, out = timed_ec(i)
        conv5(out)
        return ev

    for rnd in range(2):
        run("EdgeConv after itself", b_alone)
        run("after kNN, reading the indices that kNN launch wrote", b_knn_fresh)
        run("after kNN, reading OLD indices (kNN wrote another buffer)", b_knn_other)
        run("after conv5", b_conv5_then)
        run("kNN -> EdgeConv(old indices) -> conv5(its output)", b_step_clean)
        run("kNN -> EdgeConv(fresh indices) -> conv5(its output): the step", b_step_real)
