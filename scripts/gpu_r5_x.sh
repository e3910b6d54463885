#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( for s in 0 1 2; do timeout 900 python tools/fuzz_models.py $s 10 2>&1 | tail -3; done ) > gpurun_out/r5x_fuzz_models.txt 2>&1; cat gpurun_out/r5x_fuzz_models.txt
( for s in 0 1; do timeout 900 python tools/fuzz_gpu.py $s 2>&1 | tail -3; done ) > gpurun_out/r5x_fuzz_gpu.txt 2>&1; cat gpurun_out/r5x_fuzz_gpu.txt
