#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "set_abstraction or flownet" 2>&1 | tail -4 | cut -c1-250
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r4_bench_c5_x.json 2>gpurun_out/r4_bench_c5_x.err
python - <<P
import json
j=json.loads(open("gpurun_out/r4_bench_c5_x.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["serial_ms_per_step"], j["kernels"], j["roofline"]["frac"])
P
