#!/bin/bash
# round 3, call H: whole GPU suite + smoke + bench at the driver's arguments
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/r3h_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3h_tests.log
timeout 120 python __graft_entry__.py --smoke > gpurun_out/r3h_smoke.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3h_bench.log 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3h_tests.log | tail -40; tail -2 gpurun_out/r3h_smoke.log; tail -1 gpurun_out/r3h_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'], d['roofline']['frac'])"
