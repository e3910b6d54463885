#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/attention_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_attention_bench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "attention or dcp or pointer or transformer" 2>&1 | tail -3 | cut -c1-250
timeout 300 python tools/dcp_kernels.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/r4_dcp_kernels.txt; grep "attention\|Self CUDA time" gpurun_out/r4_dcp_kernels.txt | cut -c1-220
