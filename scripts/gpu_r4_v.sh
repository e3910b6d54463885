#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_kernels.py -m gpu -q -x -k "ball or c5 or config5" 2>&1 | tail -4 | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bq -o bq -- python $R/tools/bq_bench.py > $R/gpurun_out/r4_bq_bench.txt 2>&1 )
grep "cells=" gpurun_out/r4_bq_bench.txt
f=$(find gpurun_out/prof_bq -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200
