#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box by scripts/gpu_r3_final.sh) -> profiles/round3_* (tracked)
set -u
R=/root/repo; cd $R
cp_if() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cp_if "$(find gpurun_out/prof_r3 -name '*kernel_stats.csv' | head -1)" profiles/round3_kernel_stats_bench.csv
cp_if "$(find gpurun_out/prof_r3_c5 -name '*kernel_stats.csv' | head -1)" profiles/round3_kernel_stats_bench_c5.csv
cp_if gpurun_out/r3_bench.json profiles/round3_bench.json
cp_if gpurun_out/r3_bench_driver.json profiles/round3_bench_driver_args.json
cp_if gpurun_out/r3_bench_c5.json profiles/round3_bench_c5.json
cp_if gpurun_out/pmc_edgeconv_f16b.txt profiles/round3_pmc_edgeconv_f16b.txt
cp_if gpurun_out/pmc_conv5_f16_2p.txt profiles/round3_pmc_conv5_f16_2p.txt
cp_if gpurun_out/pmc_knn_mfma.txt profiles/round3_pmc_knn_mfma.txt
cp_if gpurun_out/pmc_group_c5.txt profiles/round3_pmc_group_c5.txt
cp_if gpurun_out/r3_kbench.txt profiles/round3_kbench.txt
python tools/kernel_meta.py > profiles/round3_kernel_resources.txt 2>/dev/null && echo "  profiles/round3_kernel_resources.txt"
python tools/traffic_json.py 3 > /dev/null && echo "  profiles/round3_traffic.json"
