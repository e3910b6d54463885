#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python tools/lab_diff.py mix l1 > gpurun_out/r4_lab_diff.txt 2>&1
cat gpurun_out/r4_lab_diff.txt
