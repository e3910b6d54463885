#!/bin/bash
# round 2, call A: the whole GPU suite (with durations), smoke, bench, rocprof kernel trace of the bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
lscpu | grep -E "Model name|^CPU\(s\)" > gpurun_out/gpu.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 -s 2>&1 | tail -250 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r2a -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit: $?" >> $R/gpurun_out/rocprof.log
cd $R
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-600
