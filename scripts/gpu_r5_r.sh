#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python tools/dcp_kernels.py > gpurun_out/r5r_dcp_kernels.txt 2>&1; grep -v "^\[W\|Warning\|_warn" gpurun_out/r5r_dcp_kernels.txt | cut -c1-90,130-200 | head -34
timeout 900 python -m pytest tests -m gpu -x -q -k "dcp or svd or transformer or attention or integration" 2>&1 | tail -3
