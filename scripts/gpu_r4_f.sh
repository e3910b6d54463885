#!/bin/bash
# round 4, call F: GPU suite after the FPS tie rule / guard changes / mix split; bench lines c2 (driver arguments) and c5
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r4_pytest_f.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_f.json 2> gpurun_out/r4_bench_f.err
tail -12 gpurun_out/r4_pytest_f.txt; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_f.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["kernels"], d["config"]["untimed_settle_probes_of_20_steps"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d.get("other_configs"))
PY
tail -3 gpurun_out/r4_bench_f.err
