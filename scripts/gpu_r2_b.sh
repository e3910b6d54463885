#!/bin/bash
# round 2, call B: f16x2 EdgeConv -- range-flag probe first (own process, short timeout), then accuracy tests, then timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -s -k "edgeconv or dgcnn" 2>&1 | tail -60 > gpurun_out/ec_tests.log
timeout 300 python tools/ec_bench.py > gpurun_out/ec_bench.log 2>&1
cat gpurun_out/ec_tests.log | tail -40; cat gpurun_out/ec_bench.log
