#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_grad_routes.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r4_pytest_n.txt
cat gpurun_out/r4_pytest_n.txt | cut -c1-250
for r in rows torch conv; do
  echo "== L3D_TRAIN_LINEAR=$r"
  L3D_TRAIN_LINEAR=$r timeout 300 python tools/train_step_profile.py --only dcp 2>&1 | grep -v amdgpu.ids | head -16 | cut -c1-160
done > gpurun_out/r4_dcp_train.txt 2>&1
cat gpurun_out/r4_dcp_train.txt
