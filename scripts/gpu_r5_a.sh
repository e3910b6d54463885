#!/bin/bash
# round 5, call a: new EMD (bits + time vs the round-4 kernel), XCD-ordered attention (time + HBM traffic)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "emd or attention or dcp" > gpurun_out/r5a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a_pytest.log
tail -5 gpurun_out/r5a_pytest.log
timeout 600 python tools/emd_bench.py > gpurun_out/r5a_emd_bench.txt 2>&1; cat gpurun_out/r5a_emd_bench.txt
timeout 300 python tools/attention_bench.py > gpurun_out/r5a_attention_bench.txt 2>&1; cat gpurun_out/r5a_attention_bench.txt
PMC_SETS="4 5" bash tools/pmc.sh attention attention_f16b_kernel > /dev/null 2>&1; cat gpurun_out/pmc_attention.txt
