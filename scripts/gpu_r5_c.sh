#!/bin/bash
# round 5, call c: matrix-core-ranked Chamfer (bits vs the exact kernel, time), EMD tests with the new backward
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chamfer_mfma.py -x -q > gpurun_out/r5c_pytest_cm.log 2>&1; tail -15 gpurun_out/r5c_pytest_cm.log
timeout 600 python tools/chamfer_bench.py > gpurun_out/r5c_chamfer_bench.txt 2>&1; cat gpurun_out/r5c_chamfer_bench.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "emd or chamfer" > gpurun_out/r5c_pytest.log 2>&1; tail -5 gpurun_out/r5c_pytest.log
