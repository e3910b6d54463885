#!/bin/bash
# round 4, call C: EdgeConv with two waves per SIMD (8 waves x 2 points, row tiles of 2 points x 8 neighbours) against the staged one-wave kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 400 python tools/variant_lab.py run ef staged w8_a0p2 w8_a1p2 w8_a0p3 w8_a0p4 > gpurun_out/r4_lab_ef_c.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_pytest_c.txt
cat gpurun_out/r4_lab_ef_c.txt; tail -8 gpurun_out/r4_pytest_c.txt
