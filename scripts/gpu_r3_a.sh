#!/bin/bash
# round 3, call A: the whole GPU suite after the route refactor (checkpointed forwards, guarded f16x2, HIP layer backward), smoke, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | tail -150 > gpurun_out/r3a_tests.log
timeout 120 python __graft_entry__.py --smoke > gpurun_out/r3a_smoke.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench.log 2>&1
tail -120 gpurun_out/r3a_tests.log; tail -3 gpurun_out/r3a_smoke.log; tail -2 gpurun_out/r3a_bench.log | cut -c1-600
