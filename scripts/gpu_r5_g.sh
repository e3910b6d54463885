#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "knn_feature or prnet or lpfa or pointconv or curvenet" > gpurun_out/r5g_pytest.log 2>&1; tail -4 gpurun_out/r5g_pytest.log
timeout 300 python tools/featknn_bench.py > gpurun_out/r5g_featknn.txt 2>&1; cat gpurun_out/r5g_featknn.txt
