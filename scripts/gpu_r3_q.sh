#!/bin/bash
# round 3, call Q: the pointer network as a channel-first pass -- tests, DCP breakdown
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad_routes.py tests/test_gpu_config_size.py -q -m gpu -p no:cacheprovider -k "transformer or dcp or DCP or layernorm or attention or hooks" 2>&1 | grep -v "^  File\|dist-packages" | tail -25
timeout 200 python tools/dcp_breakdown.py 2>&1 | grep -v amdgpu.ids | head -5
timeout 300 python tools/dcp_kernels.py 2>&1 | grep -E "^void|^[a-z_]+\(|Self CUDA time" | cut -c1-75,150-230 | head -16
