#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box) -> profiles/round2_* (tracked)
set -u
R=/root/repo; cd $R
cp_if() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cp_if "$(find gpurun_out/prof_r2_f16 -name '*kernel_stats.csv' | head -1)" profiles/round2_kernel_stats_bench.csv
cp_if "$(find gpurun_out/prof_r2_bf16x3 -name '*kernel_stats.csv' | head -1)" profiles/round2_kernel_stats_bench_bf16x3.csv
cp_if gpurun_out/r2_bench.json profiles/round2_bench.json
cp_if gpurun_out/r2_bench_driver.json profiles/round2_bench_driver_args.json
cp_if gpurun_out/pmc_edgeconv_f16.txt profiles/round2_pmc_edgeconv_f16.txt
cp_if gpurun_out/pmc_knn_mfma.txt profiles/round2_pmc_knn_mfma.txt
cp_if gpurun_out/pmc_conv5_f16.txt profiles/round2_pmc_conv5_f16.txt
cp_if gpurun_out/r2_kbench.txt profiles/round2_kbench.txt
python tools/traffic_json.py 2 > /dev/null && echo "  profiles/round2_traffic.json"
