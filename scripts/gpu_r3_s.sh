#!/bin/bash
# round 3: knn_select.hip -- parity (vs the lane kernel, vs the reference's kernel) and timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_kernels.py -q -m gpu -k "knn" -x 2>&1 | tail -15
timeout 600 python tools/knn_select_bench.py 2>&1 | tee gpurun_out/knn_select_bench.txt
