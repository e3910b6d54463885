#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X:
    clouds/sec DGCNN-fwd + Chamfer, B=32 per GPU, N=1024, k=20, emb_dims=1024 (configs[1])
    + the kNN kernel's achieved HBM GB/s against peak.

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
     127.0.0.1 --master-port P bench.py --gpus N ... -- or plainly as above: with WORLD_SIZE unset and N>1 the script
     re-launches itself under torch.distributed.run on 127.0.0.1 with a free port, one rank per GPU.)

A "step" is one pass of the hot path over one batch of synthetic clouds already resident in HBM:
DGCNN(emb_dims=1024).eval() forward on x[32,1024,3] (fused kNN -> fused EdgeConv stack -> conv5 GEMM,
all hand-written HIP) followed by ChamferDistanceLoss()(a[32,1024,3], b[32,1024,3]) including, for
N>1, the all_gather of the per-shard loss partial sums (weak scaling: 32 clouds per GPU).
Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant kernel:
the EdgeConv MFMA kernel; live HIP-event timing) and `cpu_baseline` (the reference's own CPU
op sequence -- matmul/topk kNN, torch convs, torch-fallback Chamfer -- on this box's host cores, on rank 0's tensors; N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, NPTS, KNN, EMB = 32, 1024, 20, 1024
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3         # MI355X_MICROARCH.md: dense fp32 MFMA peak
MFMA_BF16_PEAK_TF = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak (2.17 PF sustained in tools/probe_mfma_bf16.hip)
FORK_DEFAULT = "knn"              # see --fork
PRECONDITION_STEPS = 150         # untimed, before the --warmup steps (~80 ms of GPU work)
SPLIT_PRODUCTS = {"f16x2": 3, "bf16x3": 6}    # low-precision MFMA products per fp32 product
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9     # 78.6 T lane-ops/s: 256 CUs x 4 SIMD-32 x 2.4 GHz, an fma counts once (= 157.3 TFLOP/s vector fp32; SURVEY.md 8(d)'s denominator)
KNN_LANEOPS_PER_PAIR = 7         # 2 fma + 1 mul + 2 sub + compare/insert
CHAMFER_LANEOPS_PER_PAIR = 8
EMD_LANEOPS_PER_PAIR = 490       # emd.cuh:7-185: 10 levels x 3 passes x ~16.3 (8 d2 + 2 scale + 4 for v_exp_f32 at quarter rate + 1-2 weight + 1 add), DESIGN.md
# algorithmic work per cloud (restated in DESIGN.md)
KNN_BYTES_PER_CLOUD = NPTS * 3 * 4 + NPTS * KNN * 8                 # 176 128 B
CHAMFER_BYTES_PER_CLOUD = 2 * NPTS * 3 * 4 + 2 * NPTS * (4 + 4)     # 40 960 B
EDGECONV_FLOP_PER_CLOUD = NPTS * KNN * 2 * (6 * 64 + 64 * 64 + 64 * 128 + 128 * 256)
CONV5_FLOP_PER_CLOUD = NPTS * 2 * 512 * EMB


def flush_c_stdio():
    """librccl announces itself through C stdio ("Librccl path : ..."); with stdout a pipe the line sits in libc's buffer until the
    process exits and would land BEHIND rank 0's JSON line (seen on a one-rank RCCL run).  Every rank pushes it out before the
    closing barrier; rank 0 prints its line after that barrier, so it is the last line of the job's stdout."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def finish(out, rank, multi, local, dist):
    """closing barrier, then rank 0's ONE JSON line, then the process group goes"""
    flush_c_stdio()
    if multi:
        dist.barrier(device_ids=[local])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def pmc_record(tag):
    """What the committed rocprofv3 PMC passes measured for one kernel (profiles/round2_traffic.json, produced on the
    GPU box by tools/pmc.sh + tools/traffic_json.py; bench.py cannot run rocprofv3 on itself, so this is the measured
    figure of the same kernel at the same shapes).  {} if not collected."""
    for name in ("round6_traffic.json", "round5_traffic.json", "round4_traffic.json", "round3_traffic.json", "round2_traffic.json", "round1_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f).get(tag)
            if rec:
                return rec
        except (OSError, ValueError):
            pass
    return {}


def pmc_source(tag):
    """where `traffic` comes from: the committed rocprofv3 --pmc pass of this kernel at these shapes -- NOT measured in this run"""
    src = pmc_record(tag).get("source")
    return f"profiles/{src} (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel at these shapes, tools/pmc.sh; read from the committed file, not measured in this run)" if src else None


def pmc_traffic(tag):
    """HBM bytes per launch: FETCH_SIZE doubled per the guide's gfx950 correction for wide coalesced reads, + WRITE_SIZE,
    both KB -> bytes.  None if not collected."""
    return pmc_record(tag).get("hbm_bytes_per_launch")


def knn_executed(knn_ms):
    """Executed VALU work of the kNN kernel from the PMC pass (SQ_INSTS_VALU = wave-instructions per launch; x64 lanes):
    lane-operations per candidate pair and the fraction of the fp32 VALU issue peak they occupy -- the honest
    utilisation figure beside `valu_frac`, which prices only the ~7 algorithmic lane-ops per pair."""
    insts = (pmc_record("knn_mfma") or pmc_record("knn")).get("SQ_INSTS_VALU")
    if not insts:
        return {}
    laneops = insts * 64.0
    return {"executed_valu_insts_per_launch": insts,
            "executed_laneops_per_pair": laneops / (B_PER_GPU * NPTS * NPTS),
            "executed_valu_frac": laneops / (knn_ms * 1e-3) / VALU_PEAK_LANEOPS}


def edgeconv_roofline(ec_tf, ec_ms, arith):
    """MFMA roofline of the EdgeConv kernel on ALGORITHMIC fp32 flops.  fp32-MFMA kernel: peak = the dense fp32 MFMA
    rate.  Matrix-core split kernels execute every algorithmic product as P low-precision MFMA products (f16x2: 3 fp16,
    bf16x3: 6 bf16), so the ceiling for algorithmic flops is the dense fp16 / bf16 MFMA peak / P."""
    alg = B_PER_GPU * EDGECONV_FLOP_PER_CLOUD
    if arith == "fp32":
        return {"kernel": "edgeconv2_kernel<5>", "bound": "mfma", "achieved": ec_tf, "peak": MFMA_F32_PEAK_TF,
                "unit": "TFLOP/s", "frac": ec_tf / MFMA_F32_PEAK_TF, "traffic": pmc_traffic("edgeconv"), "traffic_source": pmc_source("edgeconv"),
                "avg_launch_ms": ec_ms, "algorithmic_flop_per_launch": alg}
    prods = SPLIT_PRODUCTS[arith]
    peak = MFMA_BF16_PEAK_TF / prods
    l1 = B_PER_GPU * NPTS * KNN * 2 * 6 * 64                       # layer 1 stays on the fp32 MFMA
    name = "edgeconv_f16b_kernel<5,true>" if arith == "f16x2" else "edgeconv_split_kernel<5>"
    return {"kernel": name, "bound": "mfma", "achieved": ec_tf, "peak": peak,
            "unit": "TFLOP/s", "frac": ec_tf / peak, "traffic": pmc_traffic("edgeconv_f16b" if arith == "f16x2" else "edgeconv_split") or pmc_traffic("edgeconv_f16"),
            "traffic_source": pmc_source("edgeconv_f16b" if arith == "f16x2" else "edgeconv_split") or pmc_source("edgeconv_f16"),
            "avg_launch_ms": ec_ms, "algorithmic_flop_per_launch": alg,
            "peak_note": f"fp32-equivalent ceiling = dense fp16/bf16 MFMA peak 2500 TFLOP/s / {prods} products per fp32 product "
                         "(the fp32 MFMA peak is 157.3 TFLOP/s)",
            "executed_lowprec_tflops": prods * (alg - l1) / (ec_ms * 1e-3) / 1e12,
            "lowprec_dense_peak": MFMA_BF16_PEAK_TF}


def mfma_sustained(dev, iters=6000):
    """What the matrix pipe sustains on THIS box, measured live (l3d_probe_mfma_sustained, probe.hip): every SIMD issues fp16 MFMAs back
    to back on random operands for ~1 ms.  The data-sheet peak (2.5 PFLOP/s) assumes the 2.4 GHz shader clock; under this load the chip
    holds ~1.5 GHz.  Returns {pflops, shader_mhz}; rank 0, N = 1, outside the timed region."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    sink = torch.empty(256 * 512, dtype=torch.float32, device=dev)
    ticks = torch.zeros(2, dtype=torch.int64, device=dev)
    best = None
    for rep in range(3):                                    # first launch: warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().l3d_probe_mfma_sustained(iters, ptr(sink), ptr(ticks), stream_ptr()), "l3d_probe_mfma_sustained")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = ticks.cpu().tolist()
        rec = {"pflops": iters * 4 * 32768.0 * 256 * 8 / (ms * 1e-3) / 1e15, "shader_mhz": 100.0 * t[0] / max(t[1], 1), "launch_ms": ms}
        if rep and (best is None or rec["pflops"] > best["pflops"]):
            best = rec
    return best


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_clouds=B_PER_GPU, repeats=2, seed=1000):
    """The reference's own CPU op sequence, timed on this box's host cores: oracle.dgcnn_forward_refops (matmul + topk kNN,
    utils/model_common_utils.py:3-9; the row-gather / repeat / cat graph feature, :132-156; Conv2d / BatchNorm2d / ReLU / max
    as models/dgcnn.py:34-48) + oracle.chamfer_loss_refops (the torch fallback of losses/chamfer_distance.py:5-31), on the
    SAME seeded tensors and weights the GPU side of rank 0 runs (generator seed 1000, torch.manual_seed(1) for the weights).
    The scalar C restatements (oracle.knn, oracle.chamfer_loss) stay what they are: the checker."""
    import oracle
    ncores = os.cpu_count() or 1
    torch.manual_seed(1)
    from learning3d_amd.models import DGCNN
    net = DGCNN(emb_dims=EMB).eval()
    w = {k: v.numpy() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((B_PER_GPU, NPTS, 3), generator=g)[:sample_clouds].numpy()
    a = torch.rand((B_PER_GPU, NPTS, 3), generator=g)[:sample_clouds].numpy()
    b = torch.rand((B_PER_GPU, NPTS, 3), generator=g)[:sample_clouds].numpy()
    loss = [None]

    def run(xs, as_, bs):
        t0 = time.perf_counter()
        oracle.dgcnn_forward_refops(xs, w, k=KNN)
        loss[0] = float(oracle.chamfer_loss_refops(as_, bs))
        return time.perf_counter() - t0

    with torch.no_grad():
        # the torch thread count that serves the reference's CPU path best on this host, all logical CPUs included
        # (on a 256-thread host the small conv / BN ops oversubscribe at 256; the reference user would set OMP_NUM_THREADS too)
        tried = {}
        for nthr in sorted({ncores, max(1, ncores // 2), max(1, ncores // 4), 64, 32, 16, 8} & set(range(1, ncores + 1)), reverse=True):
            torch.set_num_threads(nthr)
            run(x[:4], a[:4], b[:4])
            tried[nthr] = run(x[:4], a[:4], b[:4])
        nthr = min(tried, key=tried.get)
        torch.set_num_threads(nthr)
        best = float("inf")
        for _ in range(1 + repeats):              # first iteration is the warm-up
            best = min(best, run(x, a, b))
    return {"value": sample_clouds / best, "unit": "clouds/s", "cores": nthr, "kind": "reference-op-sequence",
            "host_cpus": ncores, "cpu_model": cpu_model(), "loss": loss[0],
            "sample": f"{sample_clouds} clouds x N={NPTS}, the tensors and weights of rank 0's GPU step (seed {seed}): the reference's "
                      f"op sequence on torch CPU -- matmul+topk kNN, gather/repeat/cat graph feature, Conv2d/BatchNorm2d/ReLU/max, "
                      f"conv5 (emb={EMB}); Chamfer = its torch fallback (broadcast difference [B,N,N,3], min, sqrt, mean); "
                      f"min of {1 + repeats} runs, torch threads={nthr} (best of {sorted(tried)} on a 4-cloud probe; host has "
                      f"{ncores} logical CPUs)"}


def other_configs(dev, iters=10, warm=3):
    """UNTIMED companion of the c2 line (N = 1 only; not part of `value`): BASELINE.json's configs[2..4] at their own sizes, a few
    steps each between two HIP events on torch's current stream (every launch of these models goes out on it), so that the
    driver's one line also carries a driver-run figure for them.  Synthetic inputs, random-init weights, eval, no_grad, eager
    launches (no hipGraph: these are end-to-end model forwards, launch overhead included)."""
    from learning3d_amd.losses import ChamferDistanceLoss
    from learning3d_amd.models import DCP, DGCNN, PCN, PointNetSetAbstraction
    from learning3d_amd.models.flownet3d import FlowNet3D

    def ms(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    out = {"note": f"untimed extras, {iters} eager steps each after {warm} warm-up steps, HIP events around the batch of steps; "
                   "not part of `value`"}
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        try:        # configs[2]: DCP-v2 registration forward (DGCNN embed + Transformer + batched 3x3 SVD head), B 32, N 1024
            torch.manual_seed(2)
            dcp = DCP(feature_model=DGCNN(emb_dims=512), pointer_="transformer", head="svd").to(dev).eval()
            tpl = (torch.rand((32, 1024, 3), generator=g) - 0.5).to(dev)
            src = (torch.rand((32, 1024, 3), generator=g) - 0.5).to(dev)
            t = ms(lambda: dcp(tpl, src))
            out["c3_dcp_v2_forward"] = {"ms_per_step": t, "clouds_per_s": 32 / t * 1e3, "shape": "B=32 pairs, N=1024, emb 512, 4 heads"}
            del dcp
        except Exception as exc:                                 # an extra must never cost the headline line
            out["c3_dcp_v2_forward"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:        # configs[3]: PCN completion forward + Chamfer loss, B 64, partial 2048 -> dense 16384
            torch.manual_seed(3)
            pcn = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).to(dev).eval()
            part = (torch.rand((64, 2048, 3), generator=g) - 0.5).to(dev)
            gt = (torch.rand((64, 16384, 3), generator=g) - 0.5).to(dev)
            cdl = ChamferDistanceLoss()
            t_f = ms(lambda: pcn(part))
            t = ms(lambda: cdl(gt, pcn(part)["fine_output"]))
            out["c4_pcn_forward_chamfer"] = {"ms_per_step": t, "clouds_per_s": 64 / t * 1e3, "pcn_forward_ms": t_f,
                                             "shape": "B=64, partial 2048 -> coarse 1024 -> fine 16384; Chamfer 16384 x 16384 per cloud"}
            del pcn, gt
        except Exception as exc:
            out["c4_pcn_forward_chamfer"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:        # configs[4] per GPU: FlowNet3D sa1 set-conv and the whole FlowNet3D forward, B 32, N 8192
            torch.manual_seed(1)
            sa = PointNetSetAbstraction(npoint=C5_S, radius=C5_R, nsample=C5_K, in_channel=3, mlp=list(C5_MLP), group_all=False).to(dev).eval()
            xyz = torch.clamp(torch.randn((32, 3, C5_N), generator=g), -2, 2).to(dev)
            feat = torch.rand((32, 3, C5_N), generator=g).to(dev)
            t = ms(lambda: sa(xyz, feat))
            out["c5_sa1_setconv"] = {"ms_per_step": t, "clouds_per_s": 32 / t * 1e3, "shape": "32 clouds per GPU, N=8192 -> 1024, r=0.5, K=16, mlp 32/32/64"}
            torch.manual_seed(4)
            fn3 = FlowNet3D().to(dev).eval()
            pc2 = (xyz + 0.05 * torch.randn((32, 3, C5_N), generator=g).to(dev)).contiguous()
            f2 = torch.rand((32, 3, C5_N), generator=g).to(dev)
            t = ms(lambda: fn3(xyz, pc2, feat, f2))
            out["c5_flownet3d_forward"] = {"ms_per_step": t, "clouds_per_s": 32 / t * 1e3, "shape": "32 cloud pairs per GPU, N=8192"}
        except Exception as exc:
            out["c5_flownet3d"] = {"error": f"{type(exc).__name__}: {exc}"}
    # ---- what README / DESIGN quote for BASELINE configs but only builder runs carried until round 4
    try:            # configs[1] in the reference's own arithmetic: every GEMM on the fp32 matrix cores (what `--arith fp32` times)
        from learning3d_amd.models import _fused
        torch.manual_seed(1)
        net = DGCNN(emb_dims=EMB).to(dev).eval()
        x = torch.rand((B_PER_GPU, NPTS, 3), generator=g).to(dev)
        a = torch.rand((B_PER_GPU, NPTS, 3), generator=g).to(dev)
        b = torch.rand((B_PER_GPU, NPTS, 3), generator=g).to(dev)
        cdl = ChamferDistanceLoss()
        prev = _fused.SPLIT_BF16
        _fused.SPLIT_BF16 = False
        try:
            with torch.no_grad():
                t = ms(lambda: (net(x), cdl(a, b)))
        finally:
            _fused.SPLIT_BF16 = prev
        out["c2_strict_fp32_arith"] = {"ms_per_step": t, "clouds_per_s": B_PER_GPU / t * 1e3,
                                       "frac_of_fp32_mfma_peak": (EDGECONV_FLOP_PER_CLOUD + CONV5_FLOP_PER_CLOUD) * B_PER_GPU / (t * 1e-3) / (MFMA_F32_PEAK_TF * 1e12),
                                       "note": "eager launches; the whole step (kNN + EdgeConv + conv5 + Chamfer) over the fp32 MFMA peak"}
        del net
    except Exception as exc:
        out["c2_strict_fp32_arith"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:            # training steps on the HIP training route (forward + backward + SGD step, train-mode BatchNorm), synthetic data
        from learning3d_amd.models import DCP, DGCNN, PCN
        tr = {}

        def train_ms(net, data, loss_fn):
            opt = torch.optim.SGD(net.parameters(), lr=1e-4)

            def step():
                opt.zero_grad(set_to_none=True)
                loss_fn(net, data).backward()
                opt.step()
            with torch.enable_grad():
                return ms(step)
        torch.manual_seed(5)
        tr["dgcnn_B32_N1024_emb1024_ms"] = train_ms(DGCNN(emb_dims=EMB).to(dev).train(), torch.rand((32, 1024, 3), generator=g).to(dev),
                                                    lambda n, d: n(d).max(dim=2)[0].square().mean())
        cdl = ChamferDistanceLoss()
        gt = (torch.rand((32, 16384, 3), generator=g) - 0.5).to(dev)
        tr["pcn_chamfer_B32_2048_to_16384_ms"] = train_ms(PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).to(dev).train(),
                                                          (torch.rand((32, 2048, 3), generator=g) - 0.5).to(dev),
                                                          lambda n, d: cdl(gt, n(d)["fine_output"]))
        del gt
        tp, sr = (torch.rand((32, 1024, 3), generator=g) - 0.5).to(dev), (torch.rand((32, 1024, 3), generator=g) - 0.5).to(dev)
        tr["dcp_v2_B32_N1024_ms"] = train_ms(DCP(feature_model=DGCNN(emb_dims=512), pointer_="transformer", head="svd").to(dev).train(), (tp, sr),
                                             lambda n, d: (lambda o: o["est_R"].square().mean() + o["est_t"].square().mean())(n(d[0], d[1])))
        tr["note"] = "forward + backward + SGD step per model, train mode, eager, HIP events around the batch of steps"
        out["training_steps"] = tr
    except Exception as exc:
        out["training_steps"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:            # configs[4] as a stream of batches (`--workload c5`, its own JSON line): pipelined and serial, one subprocess each
        import subprocess
        c5 = {}
        for tag, extra in (("pipelined", []), ("serial", ["--c5-serial"])):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c5", "--gpus", "1", "--steps", "40", "--warmup", "10",
                                "--no-cpu-baseline", "--no-other-configs"] + extra, capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                j = json.loads(line[-1])
                c5[tag] = {"clouds_per_s": j["value"], "ms_per_step": j["ms_per_step"], "roofline_frac": (j.get("roofline") or {}).get("frac")}
            else:
                c5[tag] = {"error": (r.stderr or r.stdout)[-300:]}
        c5["note"] = "bench.py --workload c5 --steps 40 --warmup 10 in a child process while this one idles: FPS + ball query + fused set-abstraction layer, 32 clouds of 8192 points per step"
        out["c5_setconv_stream"] = c5
    except Exception as exc:
        out["c5_setconv_stream"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:            # north_star's other loss: approximate EMD (losses/emd.py -> emd.hip), forward and backward at B 32, n = m = 1024
        from learning3d_amd._lib import check, lib, ptr, stream_ptr
        Be, ne = 32, 1024
        a = torch.rand((Be, ne, 3), generator=g).to(dev)
        b = torch.rand((Be, ne, 3), generator=g).to(dev)
        ws = torch.empty(lib().l3d_emd_workspace_bytes(Be, ne, ne), dtype=torch.uint8, device=dev)
        match, cost = torch.empty((Be, ne, ne), device=dev), torch.empty(Be, device=dev)
        g1, g2 = torch.empty_like(a), torch.empty_like(b)
        t_f = ms(lambda: check(lib().l3d_emd_forward(ptr(a), ptr(b), Be, ne, ne, ptr(match), ptr(cost), ptr(ws), 0, stream_ptr()), "emd"))
        t_b = ms(lambda: check(lib().l3d_emd_backward(ptr(a), ptr(b), ptr(match), Be, ne, ne, ptr(g1), ptr(g2), stream_ptr()), "emd bwd"))
        pairs = Be * ne * ne
        out["emd_c2_shape"] = {
            "forward_ms": t_f, "backward_ms": t_b, "clouds_per_s_forward": Be / t_f * 1e3, "shape": "B=32, n=m=1024",
            "roofline": {"bound": "valu", "unit": "T lane-op/s (packed fp32 peak)", "peak": VALU_PEAK_LANEOPS / 1e12,
                         "achieved": pairs * EMD_LANEOPS_PER_PAIR / (t_f * 1e-3) / 1e12,
                         "frac": pairs * EMD_LANEOPS_PER_PAIR / (t_f * 1e-3) / VALU_PEAK_LANEOPS,
                         "note": f"{EMD_LANEOPS_PER_PAIR} lane-op equivalents per pair = the reference's 30 passes x (8 for d2, 2 scale, "
                                 "v_exp_f32 counted 4 at its quarter rate, 1-2 weight, 1 add); executed here: 29 sweep + 10 match "
                                 "evaluations per pair, the match matrix written once (134 MB)"},
            "backward_match_read_gbs": 2 * 4.0 * pairs / (t_b * 1e-3) / 1e9}
    except Exception as exc:
        out["emd_c2_shape"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:        # feature-space kNN (PRNet's dynamic graphs, models/prnet.py:76-97) at B 32, N 1024, k 20
        import learning3d_amd.utils as U
        gk = torch.Generator().manual_seed(11)
        fk = {}
        for C in (64, 128):
            xf = torch.randn((32, C, 1024), generator=gk).to(dev)
            with torch.no_grad():
                t = ms(lambda: U.knn(xf, 20))
            fk[f"C{C}_us"] = t * 1e3
            # matrix work per call: 2 B N^2 C fp32-equivalent FLOP, executed as 4 fp16 products' worth (1 in the bound sweep + 3 in the collecting one)
            fk[f"C{C}_fp32_equiv_tflops"] = 2.0 * 32 * 1024 * 1024 * C / (t * 1e-3) / 1e12
        fk["shape"] = "B=32, N=1024, k=20; split pass + featknn_kernel (f16x2 GEMM + threshold / collect / rank selection)"
        out["f2_feature_space_knn"] = fk
    except Exception as exc:
        out["f2_feature_space_knn"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:        # the deterministic backward of grouping at FlowNet3D's sa1 shape, beside the reference's atomic scatter
        from learning3d_amd._lib import check, lib, ptr, stream_ptr
        from learning3d_amd.utils import pointnet2_utils as P
        gs = torch.Generator().manual_seed(12)
        Bs, Cs, Ns, Ss, Ks = 32, 64, 8192, 1024, 16
        xyz = torch.clamp(torch.randn((Bs, Ns, 3), generator=gs), -2, 2).to(dev)
        idx = P.ball_query(0.5, Ks, xyz, xyz[:, :Ss].contiguous())
        gg = torch.randn((Bs, Cs, Ss, Ks), generator=gs).to(dev)
        outg = torch.empty((Bs, Cs, Ns), device=dev)
        t_a = ms(lambda: check(lib().l3d_group_points_grad(Bs, Cs, Ns, Ss, Ks, ptr(gg), ptr(idx), ptr(outg), stream_ptr()), "group_points_grad"))
        t_d = ms(lambda: P._scatter_add_det(gg.view(Bs, Cs, Ss * Ks), idx, None, Ns, 1))
        out["f3_grouping_backward_sa1"] = {"atomic_scatter_us": t_a * 1e3, "deterministic_us": t_d * 1e3,
                                           "shape": "B=32, C=64, N=8192, 1024 groups of 16 (group_points_grad_kernel's job, utils/lib/src/group_points_gpu.cu:8-28)"}
    except Exception as exc:
        out["f3_grouping_backward_sa1"] = {"error": f"{type(exc).__name__}: {exc}"}
    torch.cuda.synchronize()
    return out


def pin_rank_to_cores(local, nlocal):
    """N launcher processes on one host: give every rank its own contiguous share of the cores this job may use and cap torch's
    CPU thread pool to it.  Unpinned, eight Python launch loops (plus eight default-sized intra-op pools) migrate over the same
    cores and the slowest rank's launch jitter becomes the step time.  Returns what was done, for the JSON line."""
    info = {"cores": None, "torch_threads": None}
    try:
        avail = sorted(os.sched_getaffinity(0))
        if nlocal > 1 and len(avail) >= nlocal:
            share = len(avail) // nlocal
            mine = avail[local * share:(local + 1) * share]
            os.sched_setaffinity(0, mine)
            info["cores"] = f"{mine[0]}-{mine[-1]} ({len(mine)} of {len(avail)})"
            torch.set_num_threads(max(1, min(8, len(mine))))
        info["torch_threads"] = torch.get_num_threads()
    except (AttributeError, OSError):
        pass
    return info


def relaunch_under_torchrun(ngpus):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this very command under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 with a free port) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd)


def selftest_cpu(args, parallel, dist):
    """Control flow only (tests/test_host_cpu.py drives it with --gpus 2): launcher -> init_from_env (gloo) -> rank-specific
    partial sums -> the same all_gather + combine and the same barrier / max-over-ranks bracketing as the real run.
    No HIP kernel and no oracle is involved; the JSON says so."""
    rank, world, local = parallel.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    part = torch.tensor([10.0 * (rank + 1), 20.0 * (rank + 1), 100.0, 200.0], dtype=torch.float64)
    placement = pin_rank_to_cores(local, world)
    if args.workload == "c5":
        # --workload c5's exchange: all_gather_into_tensor of the 4-value shard digest, summed over ranks
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        tot = None
        for _ in range(args.steps):
            flat = torch.empty(world * 4, dtype=torch.float64)
            if world > 1:
                dist.all_gather_into_tensor(flat, part)
            else:
                flat.copy_(part)
            tot = flat.view(world, 4).sum(0)
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        per_rank = [t.clone() for _ in range(world)]
        if world > 1:
            dist.all_gather(per_rank, t)
        if rank == 0:
            sr = sum(range(1, world + 1))
            print(json.dumps({"metric": "SELFTEST (control flow only, not a measurement)", "workload": "c5", "n_gpus": world,
                              "steps": args.steps, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                              "dist_backend": dist.get_backend() if world > 1 else None, "digest": [float(v) for v in tot],
                              "digest_expected": [10.0 * sr, 20.0 * sr, 100.0 * world, 200.0 * world],
                              "rank0_placement": placement, "per_rank_s": [float(v) for v in per_rank]}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    pipe = parallel.PipelinedChamferLoss()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_sync = parallel.allgather_chamfer_loss(part)
        pipe.submit(part)
    loss_pipe = pipe.flush()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = [t.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, t)
    if rank == 0:
        tot = sum(range(1, world + 1))
        want = (10.0 * tot / (100.0 * world) + 20.0 * tot / (200.0 * world)) / 2.0
        print(json.dumps({"metric": "SELFTEST (control flow only, not a measurement)", "n_gpus": world, "steps": args.steps,
                          "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                          "dist_backend": dist.get_backend() if world > 1 else None,
                          "loss_sync": float(loss_sync), "loss_pipelined": float(loss_pipe), "loss_expected": want,
                          "rank0_placement": placement,
                          "per_rank_s": [float(v) for v in per_rank]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

# ----------------------------------------------------------------------------------------------------------------------
# --workload c5: BASELINE configs[4] -- FlowNet3D's first set-conv layer (models/flownet3d.py:93-123, :293), the one config
# north_star shards: B = 256 at 8 GPUs = 32 clouds per GPU, N = 8192, npoint 1024, radius 0.5, nsample 16, mlp 32/32/64.
C5_N, C5_S, C5_K, C5_R, C5_MLP = 8192, 1024, 16, 0.5, (32, 32, 64)
# SURVEY.md 8(d), grouping c5 per GPU: idx 2 097 152 + gathered reads 12 582 912 + out [32,6,1024,16] 12 582 912 B
C5_GROUP_BYTES_PER_CLOUD = C5_S * C5_K * 4 + 2 * 6 * C5_S * C5_K * 4                    # 851 968 B  (x 32 = 27 262 976)
C5_MLP_FLOP_PER_CLOUD = C5_S * C5_K * 2 * (6 * 32 + 32 * 32 + 32 * 64)


def c5_cpu_baseline(sample_clouds=4, repeats=1):
    """oracle.set_abstraction_forward_torch (C restatements of the reference's FPS / ball query / grouping kernels + torch-CPU
    conv / BatchNorm ops, the composition pinned by tests/golden/flownet3d_sa1_c5.npz) on this box's host cores."""
    import oracle
    from learning3d_amd.models import PointNetSetAbstraction
    torch.manual_seed(1)
    sa = PointNetSetAbstraction(npoint=C5_S, radius=C5_R, nsample=C5_K, in_channel=3, mlp=list(C5_MLP), group_all=False).eval()
    w = {k: v.numpy() for k, v in sa.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    xyz = torch.clamp(torch.randn((sample_clouds, 3, C5_N), generator=g), -2, 2).numpy()
    feat = torch.rand((sample_clouds, 3, C5_N), generator=g).numpy()
    nthr = min(32, os.cpu_count() or 1)
    torch.set_num_threads(nthr)
    best = float("inf")
    for _ in range(1 + repeats):
        t0 = time.perf_counter()
        oracle.set_abstraction_forward_torch(xyz, feat, w, "", C5_S, C5_R, C5_K)
        best = min(best, time.perf_counter() - t0)
    return {"value": sample_clouds / best, "unit": "clouds/s", "cores": nthr, "kind": "port", "host_cpus": os.cpu_count(),
            "cpu_model": cpu_model(),
            "sample": f"{sample_clouds} clouds x N={C5_N} (sa1: scalar C FPS / ball query / grouping restatements, one core each, + "
                      f"torch-CPU conv stack on {nthr} threads), min of {1 + repeats} runs"}


def main_c5(args, rank, world, local, dev, dist, parallel):
    multi = world > 1 or dist.is_initialized()      # a one-rank RCCL group (L3D_INIT_SINGLE_RANK=1) takes the N > 1 code path
    from learning3d_amd.models import PointNetSetAbstraction, _fused
    depth = 0 if args.c5_serial else max(1, args.c5_depth)
    pipelined = depth > 0
    # two resident input batches, consumed alternately (a stream of batches: batch i is xyzs[i % 2])
    g = torch.Generator().manual_seed(2000 + rank)
    xyzs = [torch.clamp(torch.randn((B_PER_GPU, 3, C5_N), generator=g), -2, 2).to(dev) for _ in range(2)]   # N(0,1) clipped to [-2,2]
    feats = [torch.rand((B_PER_GPU, 3, C5_N), generator=g).to(dev) for _ in range(2)]
    torch.manual_seed(1)
    sa = PointNetSetAbstraction(npoint=C5_S, radius=C5_R, nsample=C5_K, in_channel=3, mlp=list(C5_MLP), group_all=False).to(dev).eval()

    def digest(new_xyz, new_feat):
        # the shard's digest (sum, sum of squares, count, centroid checksum): where a training loop's per-shard loss partials go
        f64 = new_feat.double()
        return torch.stack([f64.sum(), (f64 * f64).sum(), f64.new_full((), float(f64.numel())), new_xyz.double().sum()])

    def compute_serial(cur):
        with torch.no_grad():
            new_xyz, new_feat = sa(xyzs[cur], feats[cur])
            return new_feat, digest(new_xyz, new_feat)

    # Pipelined mode.  Furthest point sampling is 1024 DEPENDENT rounds on one workgroup per cloud -- 32 of 256 CUs busy for ~0.9 ms,
    # three times what the rest of the batch (ball query, grouping, MLP: every CU) takes -- and it is a function of the input
    # coordinates alone.  So it is issued `depth` batches ahead, each batch's sampling on its own stream: `depth` sampling kernels
    # (depth x 32 workgroups) are resident beside the batch being computed.  F is a ring of index buffers (even length, so that a
    # slot always holds indices of the same input parity); a slot's sampling waits for the compute that last read the slot
    # (ev_done), a batch's compute waits for its slot's sampling (ev_fps).  Every step consumes one batch and issues one sampling.
    ring = depth + 1 + ((depth + 1) % 2) if pipelined else 0
    sides = [torch.cuda.Stream() for _ in range(depth)]
    ev_fps = [torch.cuda.Event() for _ in range(ring)]
    ev_done = [torch.cuda.Event() for _ in range(ring)]
    done_valid = [False] * ring
    F = []
    if pipelined:
        with torch.no_grad():
            F = [sa.sample(xyzs[r % 2]) for r in range(ring)]
        torch.cuda.synchronize()

    def issue_sampling(j):
        s_, r = sides[j % depth], j % ring
        with torch.cuda.stream(s_), torch.no_grad():
            if done_valid[r]:
                s_.wait_event(ev_done[r])
            F[r].copy_(sa.sample(xyzs[j % 2]))
            ev_fps[r].record(s_)

    def compute_from(r):
        with torch.no_grad():
            new_xyz, new_feat = sa(xyzs[r % 2], feats[r % 2], fps_idx=F[r])
            return new_feat, digest(new_xyz, new_feat)

    graphs, graph_outs = None, None
    state = {"i": 0}

    def exchange(part, cloned):
        if multi:
            flat = torch.empty(world * 4, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(flat, part.clone() if cloned else part)   # 32 B per rank over RCCL
            return flat.view(world, 4).sum(0)
        return part

    def prime():
        """the first `depth` batches' sampling (nothing to run beside yet)"""
        state["i"] = 0
        for j in range(depth):
            issue_sampling(j)

    def step():
        i = state["i"]
        state["i"] = i + 1
        if not pipelined:
            if graphs is not None:
                graphs[i % 2].replay()
                nf, part = graph_outs[i % 2]
                return nf, exchange(part, True)
            nf, part = compute_serial(i % 2)
            return nf, exchange(part, False)
        issue_sampling(i + depth)
        r = i % ring
        main = torch.cuda.current_stream()
        main.wait_event(ev_fps[r])
        if graphs is not None:
            graphs[r].replay()
            nf, part = graph_outs[r]
        else:
            nf, part = compute_from(r)
        ev_done[r].record(main)
        done_valid[r] = True
        return nf, exchange(part, graphs is not None)

    prime()
    for _ in range(20):
        step()
    if not args.no_graph:
        try:
            torch.cuda.synchronize()
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm), torch.no_grad():
                for r in range(max(ring, 2)):
                    compute_from(r) if pipelined else compute_serial(r)
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            gs, outs = [], []
            for r in range(max(ring, 2)):       # graph r: the batch whose indices sit in F[r] (serial mode: input r, sampling included)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    outs.append(compute_from(r) if pipelined else compute_serial(r))
                gs.append(g_)
            graphs, graph_outs = gs, outs
            torch.cuda.synchronize()
            done_valid = [False] * ring
            prime()
            for _ in range(6):
                step()
            torch.cuda.synchronize()
        except Exception as exc:
            graphs = None
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eager", file=sys.stderr)
    for _ in range(args.warmup):
        step()

    def sync():
        if multi:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    digest_v = None
    for i in range(args.steps):
        _, digest_v = step()
    sync()
    elapsed = time.perf_counter() - t0

    # untimed: the kernels' own durations (HIP events around the stages of a few SERIAL eager steps on one stream) and the
    # serial step itself (sampling, then the rest, back to back: one batch's latency)
    timer = _fused.StageTimer(only=("group_kernel", "fps", "ball_query", "mlp"))
    _fused.TIMER = timer
    for c in (0, 1, 0, 1):
        compute_serial(c)
    torch.cuda.synchronize()
    _fused.TIMER = None
    stage_ms = timer.mean_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for c in (0, 1):
        compute_serial(c)
    e0.record()
    for c in (0, 1) * 5:
        compute_serial(c)
    e1.record()
    torch.cuda.synchronize()
    serial_ms = e0.elapsed_time(e1) / 10

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank = [t.clone() for _ in range(world)]
    if multi:
        dist.all_gather(per_rank, t)
    per_rank = torch.stack(per_rank).cpu()
    elapsed = float(per_rank[:, 0].max())
    if rank == 0:
        fused = "group_kernel" not in stage_ms          # sa_fused.hip: gather + three layers + max in one launch (the "mlp" stage)
        mlp_tf = B_PER_GPU * C5_MLP_FLOP_PER_CLOUD / (stage_ms["mlp"] * 1e-3) / 1e12 if stage_ms.get("mlp") else None
        if fused:
            roof = {"kernel": "sa_mlp3_kernel<8,32,32,64,2>", "bound": "mfma", "achieved": mlp_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": mlp_tf / MFMA_F32_PEAK_TF if mlp_tf else None, "traffic": pmc_traffic("sa_mlp3"), "traffic_source": pmc_source("sa_mlp3"),
                    "avg_launch_ms": stage_ms.get("mlp"), "algorithmic_flop_per_launch": B_PER_GPU * C5_MLP_FLOP_PER_CLOUD,
                    # indices in, six gathered values per (centroid, neighbour), 64 maxima per centroid out
                    "algorithmic_bytes_per_launch": B_PER_GPU * (C5_S * C5_K * 4 + C5_S * C5_K * 6 * 4 + 64 * C5_S * 4),
                    "note": "the set-abstraction layer behind the ball query as one kernel (gather, 6 -> 32 -> 32 -> 64 on the fp32 MFMA, max over "
                            "K): exact fp32 fma chains, priced against the fp32 matrix peak; the step itself is bound by furthest point "
                            "sampling's 1024 dependent rounds (one workgroup per cloud) packed beside it, see kernels.fps_ms and config.pipeline"}
        else:
            grp_ms = stage_ms["group_kernel"]
            grp_gbs = B_PER_GPU * C5_GROUP_BYTES_PER_CLOUD / (grp_ms * 1e-3) / 1e9
            roof = {"kernel": "group_concat_kernel", "bound": "hbm", "achieved": grp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": grp_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("group_c5"), "traffic_source": pmc_source("group_c5"),
                    "avg_launch_ms": grp_ms, "algorithmic_bytes_per_launch": B_PER_GPU * C5_GROUP_BYTES_PER_CLOUD,
                    "note": "microseconds of the step: the step is bound by furthest point sampling's 1024 dependent rounds "
                            "(latency, one workgroup per cloud), see kernels.fps_ms"}
        out = {
            "metric": "clouds/sec FlowNet3D sa1 set-conv (FPS + ball query + grouping + shared MLP) B=32 per GPU N=8192 -- BASELINE configs[4], "
                      "NOT the headline metric (that is the default --workload c2 line)",
            "value": world * B_PER_GPU * args.steps / elapsed, "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f32 (index work int32; shared MLP on the fp32 MFMA)", "data": "synthetic",
            "config": {"workload": "configs[4] (NOT the headline config; --workload c5): FlowNet3D sa1 set-conv forward -- furthest "
                                   "point sampling 8192 -> 1024, ball query r=0.5 K=16, grouping + shared MLP 6->32->32->64 + max "
                                   "over K (one fused kernel), eval, random-init weights; 32 clouds per GPU, two resident input batches consumed alternately",
                       "global_batch": world * B_PER_GPU, "num_points": C5_N, "npoint": C5_S, "nsample": C5_K, "radius": C5_R,
                       "launch": "hipGraph replay" if graphs is not None else "eager launches",
                       "pipeline": (f"furthest point sampling (a function of the input coordinates alone; one workgroup per cloud = 32 of 256 CUs "
                                    f"for 1024 dependent rounds) is issued {depth} batch(es) ahead, each on its own stream, beside the batch "
                                    "being computed (ball query / grouping / MLP from a ring of index buffers, event-ordered); every step "
                                    "issues one batch's sampling and produces one batch's complete output; the timed region ends when the "
                                    "last issued sampling has ended too" if pipelined else "none: sampling, then the rest, on one stream"),
                       "pipeline_depth": depth,
                       "parallelism": f"batch-sharded x{world}, no data-path collective; one 32-byte all_gather of the shard digest per step"},
            "rccl_ranks": (dist.get_world_size() if multi else 1), "dist_backend": dist.get_backend() if multi else None,
            "per_rank_ms_per_step": [float(v) / args.steps * 1e3 for v in per_rank[:, 0]],
            "rank0_placement": getattr(args, "placement", None),
            "multi_gpu_note": "no N > 1 number has been measured by the builder in any round (a gpurun box has one GPU)",
            "serial_ms_per_step": serial_ms,
            "roofline": roof,
            "kernels": {"fps_ms": stage_ms.get("fps"), "ball_query_ms": stage_ms.get("ball_query"), "group_ms": stage_ms.get("group_kernel"),
                        "mlp_ms": stage_ms.get("mlp"), "mlp_tflops": mlp_tf, "set_abstraction_fused": fused,
                        "fps_pair_evals_per_s": B_PER_GPU * C5_S * C5_N / (stage_ms["fps"] * 1e-3) if stage_ms.get("fps") else None,
                        "note": "HIP events around the stages of serial eager steps after the timed region"},
            "digest": [float(v) for v in digest_v],
        }
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = c5_cpu_baseline()
    finish(out if rank == 0 else None, rank, multi, local, dist)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default, the headline): DGCNN forward + Chamfer, 32 clouds x 1024 points per GPU; "
                         "c5: BASELINE configs[4]'s sharded layer, FlowNet3D sa1 set-conv (FPS + ball query + grouping + "
                         "3-layer shared MLP), 32 clouds x 8192 points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--c5-serial", action="store_true",
                    help="--workload c5: sampling and the rest of each batch on one stream (default: batch i + 1's furthest point "
                         "sampling runs on a second stream beside batch i's ball query / grouping / MLP)")
    ap.add_argument("--c5-depth", type=int, default=3,
                    help="--workload c5: how many batches ahead furthest point sampling is issued (one stream each; 1 = only the next batch's)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the untimed configs[2..4] extras (DCP-v2 forward, PCN + Chamfer, FlowNet3D sa1 / forward) of the c2 line")
    ap.add_argument("--sync-loss", action="store_true",
                    help="N>1: make the blocking exchange the headline (default: pipelined; both are always reported)")
    ap.add_argument("--fp32-mfma", action="store_true",
                    help="run the shared-MLP GEMMs on the fp32 MFMA (157 TF peak) instead of the matrix-core split kernels")
    ap.add_argument("--arith", choices=["f16x2", "bf16x3", "fp32"], default=None,
                    help="GEMM arithmetic of the shared-MLP kernels (default f16x2: 3 fp16 MFMA products per fp32 product; "
                         "bf16x3: 6 bf16 products, full fp32 exponent range; fp32 = --fp32-mfma)")
    ap.add_argument("--strong", nargs="?", type=int, const=0, default=None, metavar="PARTS",
                    help="strong scaling, for information (SURVEY.md 8(d)): the 32 clouds of ONE batch split over the N GPUs (32 / N each) "
                         "instead of 32 per GPU; the JSON line says \"scaling\": \"strong\".  PARTS (default: --gpus) lets one GPU run "
                         "the share it would have in a PARTS-way split")
    ap.add_argument("--no-graph", action="store_true", help="issue every step's launches eagerly instead of replaying a hipGraph")
    ap.add_argument("--fork", choices=["none", "knn", "start", "edgeconv", "conv5"], default=FORK_DEFAULT,
                    help="c2: where the step's Chamfer branch (NN search + loss tail; independent of the DGCNN chain) runs: 'knn' (default) "
                         "= on a second stream beside the kNN kernel only, joined in front of EdgeConv (two branches of the replayed "
                         "hipGraph: both kernels are VALU / latency bound and the matrix kernels keep the chip to themselves; +1.3 %); "
                         "'none' = one stream, the kernels back to back; 'start' / 'edgeconv' / 'conv5' = leaves the main stream at "
                         "the start of the step / when that stage's kernel is issued and is joined at the END of the step (measured "
                         "2 % slower than one stream: the matrix kernels lose more than the branch gains)")
    ap.add_argument("--chamfer-tail", choices=["fused", "separate"], default="separate",
                    help="ChamferDistanceLoss's loss tail as its own launch behind the search (default: 0.3-0.5 %% faster inside the replayed "
                         "two-branch graph, LABLOG R6.2) or inside the search kernel's launch (l3d_chamfer_forward_loss: what the module's "
                         "no-grad forward calls -- one launch less for an eager caller)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="control-flow self test on CPU/gloo (launcher, sharding, collective, max-over-ranks, JSON): "
                         "NO kernels run and the printed line is not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus))
    args.strong = None if args.strong is None else (args.strong or args.gpus)
    if args.strong:
        global B_PER_GPU
        if B_PER_GPU % args.strong:
            raise SystemExit(f"--strong: {B_PER_GPU} clouds do not split evenly {args.strong} ways")
        B_PER_GPU //= args.strong

    from learning3d_amd import parallel
    import torch.distributed as dist
    if args.selftest_cpu:
        return selftest_cpu(args, parallel, dist)
    from learning3d_amd.models import DGCNN, _fused
    if args.fp32_mfma or args.arith == "fp32":
        _fused.SPLIT_BF16 = False
    elif args.arith:
        _fused.GEMM_ARITH = args.arith
    from learning3d_amd.losses.chamfer_distance import ChamferDistance, chamfer_partials
    # the f16x2 range verdict: the default policy waits for the forward and reads the flag before returning (a host sync per
    # call); the bench's steps are graph replays (where nothing can be waited for) plus a few eager ones, so it takes the
    # asynchronous policy and checks the flag itself after the replays and at the end of the run (check_range(sync=True))
    _fused.RANGE_POLICY = "async"

    rank, world, local = parallel.init_from_env()
    multi = world > 1 or dist.is_initialized()          # a one-rank RCCL group (L3D_INIT_SINGLE_RANK=1) takes the N > 1 code path
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    placement = pin_rank_to_cores(local, world)     # before the device index is folded: `local` is the rank's slot on this host
    local = local % torch.cuda.device_count()     # a launcher that narrows visibility leaves one device at index 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi:
        # a multi-GPU run is an RCCL run or it is not a measurement: fail before timing anything
        if dist.get_backend() != "nccl" or dist.get_world_size() != args.gpus:
            raise SystemExit(f"[bench] expected {args.gpus} RCCL ranks (backend 'nccl'), got backend {dist.get_backend()!r} with "
                             f"{dist.get_world_size()} ranks")
        probe = torch.full((1,), float(rank + 1), dtype=torch.float64, device=dev)
        dist.all_reduce(probe)                                     # one collective over RCCL before anything is timed
        if abs(float(probe) - world * (world + 1) / 2) > 0:
            raise SystemExit(f"[bench] RCCL all_reduce probe returned {float(probe)}, expected {world * (world + 1) / 2}")
    if args.workload == "c5":
        args.placement = placement
        return main_c5(args, rank, world, local, dev, dist, parallel)

    # synthetic, seeded, already resident (weak scaling: 32 clouds per GPU; rank-specific seed)
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.rand((B_PER_GPU, NPTS, 3), generator=g).to(dev)
    a = torch.rand((B_PER_GPU, NPTS, 3), generator=g).to(dev)
    b = torch.rand((B_PER_GPU, NPTS, 3), generator=g).to(dev)
    torch.manual_seed(1)
    net = DGCNN(emb_dims=EMB).to(dev).eval()
    cd = ChamferDistance()

    loss_pipe = parallel.PipelinedChamferLoss()

    from learning3d_amd.losses.chamfer_distance import chamfer_loss_local

    branch = torch.cuda.Stream() if args.fork != "none" else None

    from learning3d_amd.losses.chamfer_distance import chamfer_forward_loss

    def chamfer_branch():
        # ChamferDistanceLoss's no-grad forward: NN search of both directions + the loss tail (N > 1: the fp64 partial sums)
        with _fused.stage("chamfer"):
            if args.chamfer_tail == "separate":            # rounds 3-5: the search, then the loss kernel (A/B: LABLOG R6.2)
                d1, d2 = cd(a, b)
                return chamfer_loss_local(d1, d2) if not multi else chamfer_partials(d1, d2)
            return chamfer_forward_loss(a, b, want="partials" if multi else "loss")

    def compute(fork=True):
        """the step's kernels: knn -> edgeconv -> conv5, Chamfer NN search, and the loss tail's local part.  The Chamfer pair
        does not depend on the DGCNN chain: with --fork it is issued on a second stream from the named point of the chain and
        joined at the end of the step (chamfer_fwd_packed_kernel's 71 VGPRs / 36 KB LDS fit beside conv5's two 204-register
        waves per SIMD and 120 KB, so its VALU work runs in the matrix kernel's shadow)."""
        with torch.no_grad():
            if branch is None or not fork:
                feat = net(x)
                return feat, chamfer_branch()
            main = torch.cuda.current_stream()
            out = []

            joined = []

            def leave(name):
                if args.fork == "knn":
                    # the Chamfer pair beside kNN only (both VALU / latency bound, a third of the issue slots each), joined again
                    # in front of EdgeConv: the two matrix kernels keep the chip to themselves
                    if name == "edgeconv" and not joined:
                        main.wait_stream(branch)
                        joined.append(True)
                    return
                if name == args.fork and not out:
                    branch.wait_stream(main)
                    with torch.cuda.stream(branch):
                        out.append(chamfer_branch())

            if args.fork == "start":
                leave("start")
            if args.fork == "knn":
                branch.wait_stream(main)
                with torch.cuda.stream(branch):
                    out.append(chamfer_branch())
            prev, _fused.ON_STAGE = _fused.ON_STAGE, leave
            try:
                feat = net(x)
            finally:
                _fused.ON_STAGE = prev
            if not out or (args.fork == "knn" and not joined):
                raise SystemExit(f"[bench] --fork {args.fork}: the DGCNN forward never entered that stage")
            main.wait_stream(branch)
        return feat, out[0]

    # The five launches of a step (four with --chamfer-tail fused) are captured once into a hipGraph and replayed: at ~0.36 ms of GPU work per step the
    # Python / ctypes launch path (~13 us per launch) had become part of the step time (0.436 ms eager).  Steps that
    # carry the live kernel-timing events (<= 8 per run) are issued eagerly.
    graph = None

    def step(sync_loss, eager=False):
        if graph is not None and not eager:
            graph.replay()
            feat, part = graph_out
        else:
            feat, part = compute(fork=not eager)           # event-carrying steps stay on one stream: clean per-kernel durations
        if not multi:
            return feat, part                              # the whole loss tail ran inside the search's launch (l3d_chamfer_forward_loss)
        if sync_loss:
            loss = parallel.allgather_chamfer_loss(part)   # blocking RCCL all_gather
        else:
            # async all_gather; returns the previous step's loss.  Under graph replay `part` is the graph's static output
            # buffer, which the next replay rewrites while this gather may still be in flight: hand the collective a copy
            loss = loss_pipe.submit(part.clone() if graph is not None and not eager else part)
        return feat, loss

    # clock / cache pre-conditioning before the W official warm-up steps: the first ~50 ms after an idle
    # period run at ramping clocks (measured: 0.545 ms/step over steps 10-60, 0.506 ms/step in steady state)
    for _ in range(PRECONDITION_STEPS):
        step(args.sync_loss)
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    compute()
            torch.cuda.current_stream().wait_stream(side)
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                graph_out = compute()
            graph = g_
            for _ in range(20):
                step(args.sync_loss)
            torch.cuda.synchronize()
            _fused.check_range(sync=True)
        except Exception as exc:                               # capture unsupported here: eager launches
            graph = None
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eager", file=sys.stderr)
    # Settle (untimed, bounded, declared in config).  On shared nodes a box now and then runs the replayed step 20-60 % slow
    # for a few hundred milliseconds although every kernel keeps its own duration (two of nine default runs on the last day
    # of round 2).  The K timed steps should measure the steady state, not that transient: a few eager steps give the sum
    # of the kernels' own durations (HIP events around each stage), then probes of 20 replayed steps run until one comes
    # within 12 % of that sum (at most 4 probes, 0.25 s apart).
    # N > 1: every rank runs the same probes (a step contains a collective, so the ranks must agree on how many steps they run):
    # the decision to probe again is itself an all_reduce -- any rank still unsettled keeps all of them probing.
    settle_probes = 0
    probe_timer = _fused.StageTimer(only=("knn", "edgeconv_kernel", "conv5", "chamfer"))
    _fused.TIMER = probe_timer
    for _ in range(3):
        step(args.sync_loss, eager=True)
    torch.cuda.synchronize()
    _fused.TIMER = None
    kernel_ms = probe_timer.mean_ms()                          # this rank's kernels, each between its own pair of HIP events
    kernel_sum_ms = sum(kernel_ms.values())
    while settle_probes < 4:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step(args.sync_loss)
        e1.record()
        torch.cuda.synchronize()
        settle_probes += 1
        unsettled = torch.tensor([1.0 if e0.elapsed_time(e1) / 20 > 1.12 * kernel_sum_ms else 0.0], device=dev)
        if multi:
            dist.all_reduce(unsettled, op=dist.ReduceOp.MAX)
        if float(unsettled) == 0.0:
            break
        time.sleep(0.25)
    for _ in range(args.warmup):
        step(args.sync_loss)

    def sync():
        if multi:
            dist.barrier(device_ids=[local])           # RCCL barrier on this rank's own device
        torch.cuda.synchronize()

    def timed(sync_loss, timer=None):
        """K steps between two (barrier + device sync) brackets; returns (this rank's seconds, last loss)."""
        # steps that carry the live kernel-timing events are issued eagerly (~0.2 ms of extra host time each): at most 8
        # per run and at most one in twenty, so a 20-step run has 1 -- in the MIDDLE of its stride, where the host is several
        # queued replays ahead of the GPU and the eager step's launch work hides behind them (as the first step after the
        # opening synchronize it stood in the open: 2-3 % of a 20-step run)
        nsamp = max(1, min(8, args.steps // 20))
        stride = max(1, -(-args.steps // nsamp))
        sync()
        t0 = time.perf_counter()
        loss = None
        for i in range(args.steps):
            sampled = timer is not None and i % stride == stride // 2
            if timer is not None:
                timer.enabled = sampled
            _, loss = step(sync_loss, eager=sampled)
        if not sync_loss:
            last = loss_pipe.flush()                      # the last step's loss, still inside the timed region
            loss = last if last is not None else loss
        sync()
        return time.perf_counter() - t0, loss

    # Live HIP-event timing of the dominant kernel inside the timed region, on the stream it is
    # launched on (torch's current stream).  Only <= 8 evenly spaced steps carry events (they are issued eagerly,
    # the others replay the graph).
    timer = _fused.StageTimer(only=("edgeconv_kernel",))
    _fused.TIMER = timer
    elapsed, loss = timed(args.sync_loss, timer)          # THE timed region: `value` comes from this one
    _fused.TIMER = None
    # N>1: the other exchange mode over the same K steps, reported beside the headline (the pipelined mode hides
    # exactly the collective latency a scaling curve is meant to show; --sync-loss swaps which one is `value`)
    other = None
    if multi:
        for _ in range(args.warmup):
            step(not args.sync_loss)
        if args.sync_loss:
            loss_pipe.flush()
        other, _ = timed(not args.sync_loss)
    stage_ms = timer.mean_ms()
    if "edgeconv_kernel" in stage_ms:                     # the live figure of the dominant kernel (events around its launch)
        stage_ms["edgeconv"] = stage_ms.pop("edgeconv_kernel")
    # untimed diagnostics (reported under "kernels" / "roofline_knn", not part of `value`): the short
    # kernels are timed as 20 back-to-back launches between two events -- a per-launch event pair adds
    # ~15 us of its own to a 25-70 us kernel.

    def per_launch_ms(fn, iters=20, reps=3):
        for _ in range(3):
            fn()
        best = None
        for _ in range(reps):                              # best of three batches: a one-off stall (allocator growth, a
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # clock step) once put 40 ms into one
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / iters
            best = t if best is None else min(best, t)
        return best

    import learning3d_amd.utils as U
    with torch.no_grad():
        xt = x.permute(0, 2, 1)
        stage_ms["knn"] = per_launch_ms(lambda: U.knn(xt, KNN))
        stage_ms["chamfer"] = per_launch_ms(lambda: cd(a, b))
        idx_ = U.knn(xt, KNN)
        packed_ = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        arith_ = _fused.gemm_arith()
        w5_, s5_, b5_, w5s_, w5f_ = net._conv5_folded()
        if arith_ == "f16x2":
            v2_ = net._packed.v2_ok
            img_ = _fused.edgeconv_forward(x, idx_, packed_, planes=True, v2=v2_, unscaled=v2_)
            stage_ms["conv5"] = per_launch_ms(lambda: _fused.pointwise_conv_f16(img_, B_PER_GPU, NPTS, w5f_, 512, EMB, s5_, b5_, relu=True,
                                                                                  unscaled=v2_))
        else:
            pooled_ = _fused.edgeconv_forward(x, idx_, packed_)
            stage_ms["conv5"] = per_launch_ms(lambda: _fused.pointwise_conv(pooled_, w5_, s5_, b5_, relu=True, channel_last=True,
                                                                              w_split=w5s_))
        if "edgeconv" not in stage_ms:
            stage_ms["edgeconv"] = per_launch_ms(lambda: _fused.edgeconv_forward(x, idx_, packed_, planes=(arith_ == "f16x2"), v2=net._packed.v2_ok,
                                                                                   unscaled=(arith_ == "f16x2" and net._packed.v2_ok)))
        _fused.check_range(sync=True)                 # no activation left the fp16 range during the run

    # max over ranks (the contract), and every rank's own time for the record
    t = torch.tensor([elapsed, other if other is not None else 0.0, kernel_ms.get("knn", 0.0), kernel_ms.get("edgeconv_kernel", 0.0),
                      kernel_ms.get("conv5", 0.0), kernel_ms.get("chamfer", 0.0)], dtype=torch.float64, device=dev)
    per_rank = [t.clone() for _ in range(world)]
    if multi:
        dist.all_gather(per_rank, t)
    else:
        per_rank = [t]
    per_rank = torch.stack(per_rank).cpu()
    elapsed = float(per_rank[:, 0].max())
    other = float(per_rank[:, 1].max()) if other is not None else None
    backend = dist.get_backend() if multi else None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        clouds_per_s = world * B_PER_GPU * args.steps / elapsed
        ec_s = stage_ms["edgeconv"] * 1e-3
        ec_tf = B_PER_GPU * EDGECONV_FLOP_PER_CLOUD / ec_s / 1e12
        knn_gbs = B_PER_GPU * KNN_BYTES_PER_CLOUD / (stage_ms["knn"] * 1e-3) / 1e9
        ch_gbs = B_PER_GPU * CHAMFER_BYTES_PER_CLOUD / (stage_ms["chamfer"] * 1e-3) / 1e9
        c5_tf = B_PER_GPU * CONV5_FLOP_PER_CLOUD / (stage_ms["conv5"] * 1e-3) / 1e12
        arith = _fused.gemm_arith()
        dtype = {"f16x2": "f32 (shared-MLP GEMMs as f16x2: fp32 operands carried as an fp16 high part + an fp16 residual (unscaled between the EdgeConv kernel and conv5, 2^12-scaled elsewhere), "
                          "3 fp16 MFMA products per fp32 product, f32 accumulate, fp32-level error -- tests bound it by 2x the "
                          "fp32-MFMA kernel's own error against fp64; distances/top-k/Chamfer plain f32)",
                 "bf16x3": "f32 (shared-MLP GEMMs as bf16x3: fp32 operands split exactly into 3 bf16 planes, 6 bf16 MFMA "
                           "products per fp32 product, f32 accumulate, fp32-level error; distances/top-k/Chamfer plain f32)",
                 "fp32": "f32"}[arith]
        out = {
            "metric": "clouds/sec DGCNN-fwd+Chamfer B=32 N=1024; kNN HBM GB/s vs peak at 1/2/4/8 GPU",
            "value": clouds_per_s, "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: DGCNN k=20 kNN + EdgeConv forward (emb_dims=1024, eval, random-init "
                                   "weights) + ChamferDistanceLoss, B=32 clouds per GPU, N=1024, inputs resident in HBM",
                       "global_batch": world * B_PER_GPU, "num_points": NPTS, "k": KNN, "emb_dims": EMB,
                       "untimed_precondition_steps": PRECONDITION_STEPS, "untimed_settle_probes_of_20_steps": settle_probes,
                       "settle_rule": "probes of 20 replayed steps until one is within 12 % of the kernels' own sum (max 4); for N > 1 all "
                                      "ranks probe until every rank is settled (all_reduce of the verdict)",
                       "launch": ("hipGraph replay of the step's " + ("5 kernels" if args.chamfer_tail == "separate" else "4 kernels (Chamfer search + loss tail in one)")) if graph is not None else "eager launches",
                       "chamfer_branch": ("one stream" if branch is None else
                                          "second stream beside the kNN kernel only, joined in front of EdgeConv (two branches of the replayed hipGraph)"
                                          if args.fork == "knn" else
                                          f"second stream, leaves the chain at '{args.fork}', joined at the end of the step"),
                       "parallelism": f"batch-sharded x{world}, all_gather of loss partials only"
                                      + ("" if args.sync_loss or not multi else " (asynchronous, consumed one step later)")},
            # multi-GPU record: ranks that really joined the process group, its backend ("nccl" = RCCL on ROCm),
            # each rank's own wall time for the K steps, and the OTHER loss-exchange mode timed over the same K steps
            "rccl_ranks": (dist.get_world_size() if multi else 1), "dist_backend": backend,
            "per_rank_ms_per_step": [float(v) / args.steps * 1e3 for v in per_rank[:, 0]],
            # every rank's own kernels (eager steps before the timed region, one pair of HIP events per stage): with the per-rank step
            # times above, a scaling loss splits into slowest rank vs exchange vs host
            "per_rank_kernel_ms": [{"knn": float(r[2]), "edgeconv": float(r[3]), "conv5": float(r[4]), "chamfer_and_loss_tail": float(r[5]),
                                    "sum": float(r[2] + r[3] + r[4] + r[5])} for r in per_rank],
            "rank0_placement": placement,
            "multi_gpu_note": "no N > 1 number has been measured by the builder in any round (a gpurun box has one GPU); the N > 1 code "
                              "path has run over RCCL with one rank and over gloo with two",
            "loss_exchange": ("blocking all_gather inside the step" if args.sync_loss else
                              "asynchronous all_gather, consumed one step later") if multi else "none (1 rank)",
            "other_exchange_mode": None if other is None else {
                "mode": "asynchronous all_gather, consumed one step later" if args.sync_loss else "blocking all_gather inside the step",
                "ms_per_step": other / args.steps * 1e3, "value": world * B_PER_GPU * args.steps / other},
            # dominant kernel by time: the fused 4-layer EdgeConv stack
            "roofline": edgeconv_roofline(ec_tf, stage_ms["edgeconv"], arith),
            # the metric's second half: kNN (and Chamfer) HBM rate on ALGORITHMIC bytes
            "roofline_knn": {"kernel": "knn_mfma_kernel<8>", "bound": "hbm", "achieved": knn_gbs,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": knn_gbs / HBM_PEAK_GBS,
                             "traffic": pmc_traffic("knn_mfma") or pmc_traffic("knn"),
                             "traffic_source": pmc_source("knn_mfma") or pmc_source("knn"),
                             "avg_launch_ms": stage_ms["knn"],
                             "algorithmic_bytes_per_launch": B_PER_GPU * KNN_BYTES_PER_CLOUD,
                             "note": "VALU-bound by construction (33.5 M pair evaluations per 5.6 MB; ranking values come from the fp32 matrix cores, selection is VALU work), see DESIGN.md",
                             "pair_evals_per_s": B_PER_GPU * NPTS * NPTS / (stage_ms["knn"] * 1e-3),
                             "valu_frac": B_PER_GPU * NPTS * NPTS * KNN_LANEOPS_PER_PAIR /
                             (stage_ms["knn"] * 1e-3) / VALU_PEAK_LANEOPS,
                             **knn_executed(stage_ms["knn"])},
            "kernels": {"knn_ms": stage_ms["knn"], "edgeconv_ms": stage_ms["edgeconv"], "conv5_ms": stage_ms["conv5"],
                        "chamfer_ms": stage_ms["chamfer"], "conv5_tflops": c5_tf, "chamfer_alg_gbs": ch_gbs,
                        "chamfer_valu_frac": 2 * B_PER_GPU * NPTS * NPTS * CHAMFER_LANEOPS_PER_PAIR /
                        (stage_ms["chamfer"] * 1e-3) / VALU_PEAK_LANEOPS,
                        "knn_chamfer_alg_gbs": B_PER_GPU * (KNN_BYTES_PER_CLOUD + CHAMFER_BYTES_PER_CLOUD) /
                        ((stage_ms["knn"] + stage_ms["chamfer"]) * 1e-3) / 1e9},
            "loss": float(loss),
        }
        if not multi and arith in SPLIT_PRODUCTS:
            try:        # the roofline's denominator, re-measured: the matrix rate this box sustains under an all-MFMA load
                sus = mfma_sustained(dev)
                prods = SPLIT_PRODUCTS[arith]
                out["roofline"]["sustained"] = {
                    "lowprec_pflops": sus["pflops"], "shader_mhz": sus["shader_mhz"], "probe_launch_ms": sus["launch_ms"],
                    "fp32_equiv_ceiling_tflops": sus["pflops"] * 1e3 / prods,
                    "frac_of_sustained": out["roofline"]["achieved"] / (sus["pflops"] * 1e3 / prods),
                    "note": "l3d_probe_mfma_sustained: every SIMD issuing v_mfma_f32_32x32x16_f16 back to back on random operands for ~1 ms "
                            "after the timed region; `frac` above stays priced against the data sheet's 2500 TFLOP/s dense peak, which "
                            "assumes a 2.4 GHz shader clock this load does not hold"}
            except Exception as exc:
                out["roofline"]["sustained"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not multi and not args.no_other_configs:
            out["other_configs"] = other_configs(dev)
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["loss_matches_gpu_step"] = abs(out["cpu_baseline"]["loss"] - out["loss"]) < 1e-6
    finish(out if rank == 0 else None, rank, multi, local, dist)


if __name__ == "__main__":
    main()
