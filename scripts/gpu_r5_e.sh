#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chamfer_mfma.py -x -q > gpurun_out/r5e_pytest_cm.log 2>&1; tail -3 gpurun_out/r5e_pytest_cm.log
timeout 600 python tools/chamfer_bench.py > gpurun_out/r5e_chamfer_bench.txt 2>&1; cat gpurun_out/r5e_chamfer_bench.txt
