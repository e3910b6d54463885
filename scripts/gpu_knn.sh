#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 60 tools/bin/probe_mfma_dot > gpurun_out/probe_mfma_dot.log 2>&1; cat gpurun_out/probe_mfma_dot.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "knn" > gpurun_out/knn_tests.log 2>&1; tail -15 gpurun_out/knn_tests.log
timeout 300 python tools/knn_bench.py > gpurun_out/knn_bench.log 2>&1; cat gpurun_out/knn_bench.log
