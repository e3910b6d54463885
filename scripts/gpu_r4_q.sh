#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_grad_routes.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-250
timeout 600 python tools/fuzz_gpu.py 7 2>&1 | grep "bmm\|fuzz OK\|Error\|assert" | head
for r in rows torch; do
  L3D_TRAIN_LINEAR=$r timeout 300 python tools/dcp_train_cprofile.py 3 2>&1 | grep "^route"
done > gpurun_out/r4_dcp_train_q.txt 2>&1
cat gpurun_out/r4_dcp_train_q.txt
timeout 300 python tools/train_step_profile.py --only dcp > gpurun_out/r4_dcp_train_trace.txt 2>&1; grep "==\|bmm\|Cijk" gpurun_out/r4_dcp_train_trace.txt | head -8 | cut -c1-200
