#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5_fk
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5_fk -o fk -- python $R/tools/featknn_bench.py > $R/gpurun_out/r5h_prof.log 2>&1
f=$(find $R/gpurun_out/prof_r5_fk -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -12
