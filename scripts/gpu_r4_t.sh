#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/dcp_kernels.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/r4_dcp_kernels.txt
cat gpurun_out/r4_dcp_kernels.txt | head -50
