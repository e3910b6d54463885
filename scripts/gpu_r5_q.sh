#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python tools/emd_bench.py > gpurun_out/r5q_emd_bench.txt 2>&1; cat gpurun_out/r5q_emd_bench.txt
