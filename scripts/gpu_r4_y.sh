#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for seed in 1 2 3; do timeout 600 python tools/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | tail -22 | cut -c1-200; done > gpurun_out/r4_fuzz.txt
tail -40 gpurun_out/r4_fuzz.txt
