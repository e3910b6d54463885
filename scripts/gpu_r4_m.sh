#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for d in 1 2 3 4 6; do
  timeout 200 python bench.py --workload c5 --steps 100 --warmup 10 --c5-depth $d --no-cpu-baseline > gpurun_out/r4_bench_c5_d$d.json 2>> gpurun_out/r4_bench_c5m.err
  python - <<P
import json
try:
    j=json.loads(open("gpurun_out/r4_bench_c5_d$d.json").read().strip().splitlines()[-1])
    print("depth $d", j["value"], j["ms_per_step"], j["serial_ms_per_step"], j["config"]["launch"], j["digest"])
except Exception as e: print("depth $d failed", e)
P
done
timeout 200 python bench.py --workload c5 --steps 100 --warmup 10 --c5-serial --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial', j['value'], j['ms_per_step'], j['digest'])"
tail -5 gpurun_out/r4_bench_c5m.err | grep -v amdgpu.ids
