#!/bin/bash
# round 2, call C: f16x2 EdgeConv + conv5 end to end: tests, kernel timings, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -s -k "edgeconv or dgcnn or conv_f16 or conv_split" 2>&1 | tail -60 > gpurun_out/ec_tests.log
timeout 300 python tools/ec_bench.py > gpurun_out/ec_bench.log 2>&1
timeout 300 python tools/conv5_bench.py >> gpurun_out/ec_bench.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/ec_tests.log | tail -8; grep -v amdgpu.ids gpurun_out/ec_bench.log; tail -1 gpurun_out/bench.log | cut -c1-400; python - <<'PY'
import json
try:
    j=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
    print({k:j[k] for k in ('value','ms_per_step')}, j['kernels'], j['roofline']['frac'])
except Exception as e: print('bench parse', e)
PY
