"""Times l3d_fold_mlp_f16 at config 4's shape (B 64, 16 384 fine points) against a second build of the kernel, if
tools/bin/libfold_old.so exists (an older source with the entry point renamed l3d_fold_mlp_f16_old), and compares outputs."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from learning3d_amd._lib import check, lib, ptr, stream_ptr
from learning3d_amd.models import _fused


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    g = torch.Generator().manual_seed(1)
    B, N = 64, 16384
    dev = "cuda"
    x5 = (torch.rand((B, N, 5), generator=g) - 0.5).to(dev)
    w5g = (torch.randn((512, 5), generator=g) * 0.3).to(dev)
    s5 = torch.randn((B, 512), generator=g).to(dev)
    w6 = (torch.randn((512, 512), generator=g) / 22).to(dev)
    b6 = (torch.randn(512, generator=g) * 0.1).to(dev)
    w7 = (torch.randn((3, 512), generator=g) / 22).to(dev)
    b7 = torch.randn(3, generator=g).to(dev)
    ce = torch.randn((B, N, 3), generator=g).to(dev)
    planes = _fused.split_weights_f16(w6)
    out = torch.empty((B, N, 3), device=dev)

    def run(fn, o):
        check(fn(ptr(x5), 5, ptr(w5g), ptr(s5), ptr(planes), ptr(b6), ptr(w7), ptr(b7), ptr(ce), B, N, ptr(o), stream_ptr()), "fold")
    t = timeit(lambda: run(lib().l3d_fold_mlp_f16, out))
    flops = 2.0 * B * N * 512 * 512
    print(f"l3d_fold_mlp_f16      {t:9.1f} us   {flops / t / 1e6:7.1f} TFLOP/s fp32-equivalent (conv6 only)")
    old = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libfold_old.so")
    if os.path.exists(old):
        L = ctypes.CDLL(old)
        out2 = torch.empty_like(out)
        t2 = timeit(lambda: run(L.l3d_fold_mlp_f16_old, out2))
        print(f"l3d_fold_mlp_f16_old  {t2:9.1f} us   max |new - old| {float((out - out2).abs().max()):.3e}  (|out| max {float(out.abs().max()):.2f})")
    # fp64 reference on one cloud
    h5 = torch.relu(s5[:1, None, :].double() + x5[:1].double() @ w5g.double().T)
    h6 = torch.relu(h5 @ w6.double().T + b6.double())
    ref = h6 @ w7.double().T + b7.double() + ce[:1].double()
    print(f"max |new - fp64| on cloud 0: {float((out[:1].double() - ref).abs().max()):.3e}")


if __name__ == "__main__":
    main()
