#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for s in 0 1 2 3 4 5; do timeout 600 python tools/fuzz_geometry.py $s 12 2>&1 | tail -2; done > gpurun_out/r5n_fuzz.txt 2>&1; cat gpurun_out/r5n_fuzz.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "emd or chamfer" 2>&1 | tail -3
