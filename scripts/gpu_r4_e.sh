#!/bin/bash
# round 4, call E: EdgeConv split by v_fma_mix (3 instructions per value pair) and ring depth 3; new parity tests; c5 pipelined vs serial
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 400 python tools/variant_lab.py run ef mix pd3 mixpd3 > gpurun_out/r4_lab_ef_e.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r4_pytest_e.txt
timeout 300 python bench.py --workload c5 --steps 50 --warmup 10 > gpurun_out/r4_bench_c5_pipe.json 2> gpurun_out/r4_bench_c5.err
timeout 300 python bench.py --workload c5 --steps 50 --warmup 10 --c5-serial --no-cpu-baseline > gpurun_out/r4_bench_c5_serial.json 2>> gpurun_out/r4_bench_c5.err
cat gpurun_out/r4_lab_ef_e.txt; tail -12 gpurun_out/r4_pytest_e.txt; cut -c1-1500 gpurun_out/r4_bench_c5_pipe.json; echo; cut -c1-600 gpurun_out/r4_bench_c5_serial.json; tail -5 gpurun_out/r4_bench_c5.err
