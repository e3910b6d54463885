#!/bin/bash
# round 3, call J: kNN with padded candidate blocks: bit-exactness tests, timing, LDS-conflict PMC pass; full suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3j_tests.log 2>&1
timeout 200 python tools/knn_bench.py > gpurun_out/r3j_knn.log 2>&1
bash tools/pmc.sh knn_mfma knn_mfma_kernel > /dev/null 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3j_tests.log | tail -8; cat gpurun_out/r3j_knn.log; grep -E "LDS|SQ_INSTS_VALU|GRBM_GUI" gpurun_out/pmc_knn_mfma.txt | sort -u
