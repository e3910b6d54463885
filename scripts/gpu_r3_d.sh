#!/bin/bash
# round 3, call D: bench --workload c5 (first run), kernel trace of it
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python bench.py --workload c5 --steps 50 --warmup 10 > gpurun_out/r3d_bench_c5.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3d_prof_c5 -o c5 -- python $R/bench.py --workload c5 --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $R/gpurun_out/r3d_prof_c5.log 2>&1
cd $R
f=$(find gpurun_out/r3d_prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" > gpurun_out/r3d_c5_kernel_stats.csv
tail -3 gpurun_out/r3d_bench_c5.log | cut -c1-3000; cat gpurun_out/r3d_c5_kernel_stats.csv | cut -c1-200
