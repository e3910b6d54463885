#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for d in 2 3 4 5 6; do
  timeout 200 python bench.py --workload c5 --steps 200 --warmup 20 --c5-depth $d --no-cpu-baseline > gpurun_out/r4_bench_c5_d$d.json 2>> gpurun_out/r4_bench_c5m.err
  python - <<P
import json
j=json.loads(open("gpurun_out/r4_bench_c5_d$d.json").read().strip().splitlines()[-1])
print("depth $d", round(j["value"]), round(j["ms_per_step"],4), j["digest"][:2])
P
done
