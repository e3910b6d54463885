#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python tools/ec_instep_probe.py > gpurun_out/r5t_ec_instep.txt 2>&1; cat gpurun_out/r5t_ec_instep.txt
