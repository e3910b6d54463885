#!/bin/bash
# round 3, call K: restructured attention kernel: accuracy vs fp64, DCP goldens, timing, DCP breakdown
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "attention or dcp or transformer or fused_kernels_other" > gpurun_out/r3k_tests.log 2>&1
timeout 200 python tools/attention_bench.py > gpurun_out/r3k_att.log 2>&1
timeout 300 python tools/dcp_breakdown.py > gpurun_out/r3k_dcp.log 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3k_tests.log | tail -30; cat gpurun_out/r3k_att.log; head -6 gpurun_out/r3k_dcp.log
