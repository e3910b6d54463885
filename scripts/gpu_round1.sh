#!/bin/bash
# GPU-box script: parity tests, smoke, bench, kernel microbench, rocprof kernel trace.
# usage: ./scripts_gpu_round1.sh [quick]      (quick: tests + kbench + bench only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
lscpu | grep -E "Model name|^CPU\(s\)" > gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/kbench.py > gpurun_out/kbench.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench.log
if [ "${1:-full}" != "quick" ]; then
  timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof exit: $?" >> $R/gpurun_out/rocprof.log
  cd $R
  find gpurun_out/prof -name "*stats*" | head -5
  tail -3 gpurun_out/smoke.log
fi
tail -4 gpurun_out/pytest_gpu.log; cat gpurun_out/kbench.log | head -20; tail -2 gpurun_out/bench.log | cut -c1-1500
