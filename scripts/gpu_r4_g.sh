#!/bin/bash
# round 4, call G: EdgeConv prologue by LDS-DMA, layer 1's finish under its second pair's MFMAs; fixed cost per launch (B = 64);
# conv5 with the two waves of a SIMD half a chunk apart; GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 400 python tools/variant_lab.py run ef pro l1 > gpurun_out/r4_lab_ef_g.txt 2>&1
L3D_LAB_B=64 timeout 400 python tools/variant_lab.py run ef l1 > gpurun_out/r4_lab_ef_g64.txt 2>&1
timeout 400 python tools/variant_lab.py run cf base stag > gpurun_out/r4_lab_cf_g.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4_pytest_g.txt
cat gpurun_out/r4_lab_ef_g.txt gpurun_out/r4_lab_ef_g64.txt gpurun_out/r4_lab_cf_g.txt; tail -6 gpurun_out/r4_pytest_g.txt
