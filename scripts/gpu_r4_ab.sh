#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r4_bench_ab.json 2>gpurun_out/r4_bench_ab.err
python - <<P
import json
j=json.loads(open("gpurun_out/r4_bench_ab.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["kernels"])
P
timeout 600 python tools/kbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_kbench.txt; grep -i "ball\|knn\|chamfer\|three\|density" gpurun_out/r4_kbench.txt | head -30 | cut -c1-160
