#!/bin/bash
# round 3, call F: two-plane EdgeConv after the scaling fix: accuracy + per-layer cycle breakdown of both variants
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_disk_feed.py -m gpu -q -p no:cacheprovider -s -k "edgeconv or dgcnn or resident" > gpurun_out/r3f_tests.log 2>&1
for v in 0 1; do timeout 120 tools/bin/pef_v$v > gpurun_out/r3f_pef_v$v.log 2>&1; done
grep -E "two-plane|f16x2 B|passed|failed|Error" gpurun_out/r3f_tests.log | tail -30; cat gpurun_out/r3f_pef_v0.log gpurun_out/r3f_pef_v1.log
