#!/bin/bash
# round 5, call f: whole GPU suite + the default bench line with the new other_configs entries
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r5f_pytest.log 2>&1; tail -8 gpurun_out/r5f_pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5f_bench.log 2>&1; grep "^{" gpurun_out/r5f_bench.log > gpurun_out/r5f_bench.json; tail -3 gpurun_out/r5f_bench.log | cut -c1-3000
