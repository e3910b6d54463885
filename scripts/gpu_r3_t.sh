#!/bin/bash
# round 3, call T: fold_mlp_f16 two-plane vs the previous build, whole GPU suite, kbench for profiles/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/fold_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3t_fold.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r3t_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3t_tests.log
tail -3 gpurun_out/r3t_tests.log
timeout 900 python tools/kbench.py > gpurun_out/r3_kbench.txt 2>&1
grep -v "^{" gpurun_out/r3_kbench.txt | tail -34
