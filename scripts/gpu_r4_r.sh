#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-250 > gpurun_out/r4_pytest_r.txt
cat gpurun_out/r4_pytest_r.txt
for r in rows torch; do
  L3D_TRAIN_LINEAR=$r timeout 300 python tools/dcp_train_cprofile.py 3 2>&1 | grep "^route"
done > gpurun_out/r4_dcp_train_r.txt 2>&1
cat gpurun_out/r4_dcp_train_r.txt
