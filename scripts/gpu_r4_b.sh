#!/bin/bash
# round 4, call B: conv5 variants (round-3 kernel, swapped operands + dwordx4 epilogue, same source without the swap), GPU suite, bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/variant_lab.py run cf r3 swap noswap > gpurun_out/r4_lab_cf_b.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_pytest_b.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_b.json 2> gpurun_out/r4_bench_b.err
cat gpurun_out/r4_lab_cf_b.txt; tail -5 gpurun_out/r4_pytest_b.txt; tail -c 4000 gpurun_out/r4_bench_b.json; tail -5 gpurun_out/r4_bench_b.err
