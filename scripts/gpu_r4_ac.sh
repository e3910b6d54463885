#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | cut -c1-250
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r4_bench_c5_ac.json 2>/dev/null
python - <<P
import json
j=json.loads(open("gpurun_out/r4_bench_c5_ac.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["serial_ms_per_step"], j["kernels"]["fps_ms"])
P
for r in rows torch; do L3D_TRAIN_LINEAR=$r timeout 300 python tools/dcp_train_cprofile.py 3 2>&1 | grep "^route"; done
