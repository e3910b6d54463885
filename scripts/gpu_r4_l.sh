#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/variant_lab.py run cf "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_lab_cf.txt
cat gpurun_out/r4_lab_cf.txt | cut -c1-200
