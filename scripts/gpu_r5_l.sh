#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python tools/variant_lab.py run cf "$@" > gpurun_out/r5l_lab_cf.txt 2>&1; cat gpurun_out/r5l_lab_cf.txt
