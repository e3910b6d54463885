#!/bin/bash
# round 3, call I: two-plane conv5: tests, A/B timing, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3i_tests.log 2>&1
timeout 300 python tools/conv5_bench.py > gpurun_out/r3i_c5.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3i_bench.log 2>&1
timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3i_bench2.log 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3i_tests.log | tail -25; cat gpurun_out/r3i_c5.log; for f in gpurun_out/r3i_bench.log gpurun_out/r3i_bench2.log; do tail -1 $f | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'], d['roofline']['frac'])"; done
