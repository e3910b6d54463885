#!/bin/bash
# round 4, call H: EdgeConv with the DMA prologue (wait fixed) + layer-1 overlap against the known-good fma_mix build; GPU suite; bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
L3D_LAB_REF=mix timeout 400 python tools/variant_lab.py run ef mix l1 > gpurun_out/r4_lab_ef_h.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4_pytest_h.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_h.json 2> gpurun_out/r4_bench_h.err
cat gpurun_out/r4_lab_ef_h.txt; tail -6 gpurun_out/r4_pytest_h.txt; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_h.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["kernels"])
PY
