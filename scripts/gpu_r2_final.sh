#!/bin/bash
# Round-2 evidence pass: GPU tests, smoke, bench (defaults + driver arguments), rocprofv3 kernel stats (both matrix-core
# arithmetics), PMC passes of the two f16x2 kernels, per-kernel micro-benchmarks.  Everything lands under gpurun_out/;
# scripts/collect_profiles.sh (run back in the container) copies the summaries into profiles/round2_*.
# usage: scripts/gpu_r2_final.sh [tests|notests]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
if [ "${1:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2_smoke.log 2>&1; tail -1 gpurun_out/r2_smoke.log
fi
timeout 600 python bench.py > gpurun_out/r2_bench.log 2>&1; tail -1 gpurun_out/r2_bench.log > gpurun_out/r2_bench.json; cut -c1-200 gpurun_out/r2_bench.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_driver.log 2>&1; tail -1 gpurun_out/r2_bench_driver.log > gpurun_out/r2_bench_driver.json; cut -c1-200 gpurun_out/r2_bench_driver.json
bash scripts/gpu_prof.sh r2_f16
bash scripts/gpu_prof.sh r2_bf16x3 "--arith bf16x3"
bash tools/pmc.sh edgeconv_f16 edgeconv_f16 > /dev/null 2>&1; grep -E "FETCH|WRITE|MFMA|GUI" gpurun_out/pmc_edgeconv_f16.txt
bash tools/pmc.sh knn "knn_mfma_kernel" > /dev/null 2>&1; cp gpurun_out/pmc_knn.txt gpurun_out/pmc_knn_mfma.txt; grep -E "FETCH|WRITE|INSTS_VALU |GUI" gpurun_out/pmc_knn_mfma.txt
bash tools/pmc.sh conv5_f16 "^(void )?conv_f16_kernel" > /dev/null 2>&1; grep -E "FETCH|WRITE|MFMA|GUI" gpurun_out/pmc_conv5_f16.txt
cd $R && timeout 600 python tools/kbench.py > gpurun_out/r2_kbench.txt 2>&1; tail -5 gpurun_out/r2_kbench.txt
