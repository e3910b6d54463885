#!/bin/bash
# round 4, call J: GPU suite + bench lines after the prune (90 entry points)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_pytest_j.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_j.json 2> gpurun_out/r4_bench_j.err
tail -25 gpurun_out/r4_pytest_j.txt; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_j.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"], {k: v.get("ms_per_step") for k, v in d["other_configs"].items() if isinstance(v, dict)})
PY
tail -3 gpurun_out/r4_bench_j.err
