#!/bin/bash
# round 3, call G: persistent two-plane EdgeConv: tests, A/B timing, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3g_tests.log 2>&1
timeout 300 python tools/ec_bench.py f16b f16b-planes f16 f16-planes > gpurun_out/r3g_ec.log 2>&1
timeout 120 tools/bin/pef_v1 > gpurun_out/r3g_pef_v1.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3g_bench.log 2>&1
grep -v "^  File\|dist-packages" gpurun_out/r3g_tests.log | tail -30; cat gpurun_out/r3g_ec.log gpurun_out/r3g_pef_v1.log; tail -1 gpurun_out/r3g_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'], d['roofline']['frac'])"
