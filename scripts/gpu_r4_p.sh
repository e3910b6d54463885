#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/bmm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_bmm_bench.txt
cat gpurun_out/r4_bmm_bench.txt
timeout 600 python -m pytest tests/test_gpu_grad_routes.py -m gpu -q -x -k "bmm or pointer or square or softmax" 2>&1 | tail -3 | cut -c1-250
