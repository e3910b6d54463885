#!/bin/bash
# round 3, call E: two-plane f16x2 EdgeConv kernel: accuracy tests, A/B timing, disk feed test, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_disk_feed.py tests/test_gpu_grad_routes.py -m gpu -q -p no:cacheprovider -s -k "edgeconv or dgcnn or resident or fused" > gpurun_out/r3e_tests.log 2>&1
timeout 300 python tools/ec_bench.py > gpurun_out/r3e_ec.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3e_bench.log 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3e_tests.log | tail -60; cat gpurun_out/r3e_ec.log; tail -1 gpurun_out/r3e_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'], d['roofline']['frac'])"
