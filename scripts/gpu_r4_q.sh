#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | cut -c1-250 > gpurun_out/r4_pytest_q.txt
cat gpurun_out/r4_pytest_q.txt
for r in rows torch; do
  L3D_TRAIN_LINEAR=$r timeout 300 python tools/dcp_train_cprofile.py 3 2>&1 | grep "^route"
done > gpurun_out/r4_dcp_train_q.txt 2>&1
cat gpurun_out/r4_dcp_train_q.txt
timeout 200 python bench.py --workload c5 --steps 100 --warmup 10 --c5-depth 2 --no-cpu-baseline > gpurun_out/r4_bench_c5_q.json 2>/dev/null; cut -c1-300 gpurun_out/r4_bench_c5_q.json
