#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "edgeconv" 2>&1 | tail -40 > gpurun_out/r4_pytest_k.txt
cat gpurun_out/r4_pytest_k.txt | cut -c1-220
