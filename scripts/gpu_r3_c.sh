#!/bin/bash
# round 3, call C: whole GPU suite (driver's command), then bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12 > gpurun_out/r3c_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3c_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3c_bench.log 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3c_tests.log | tail -120; tail -2 gpurun_out/r3c_bench.log | cut -c1-400
