#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
L3D_TRAIN_LINEAR=rows timeout 300 python tools/dcp_train_cprofile.py 30 2>&1 | grep -v amdgpu.ids | cut -c1-180 > gpurun_out/r4_dcp_cprof.txt
L3D_TRAIN_LINEAR=torch timeout 300 python tools/dcp_train_cprofile.py 12 2>&1 | grep -v amdgpu.ids | cut -c1-180 >> gpurun_out/r4_dcp_cprof.txt
cat gpurun_out/r4_dcp_cprof.txt
