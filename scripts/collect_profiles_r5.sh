#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box by scripts/gpu_r5_final.sh) -> profiles/round5_* (tracked)
set -u
R=/root/repo; cd $R
cp_if() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cp_if "$(find gpurun_out/prof_r5 -name '*kernel_stats.csv' | head -1)" profiles/round5_kernel_stats_bench.csv
cp_if "$(find gpurun_out/prof_r5_emd -name '*kernel_stats.csv' | head -1)" profiles/round5_kernel_stats_emd.csv
cp_if gpurun_out/r5_bench.json profiles/round5_bench.json
cp_if gpurun_out/r5_bench_driver.json profiles/round5_bench_driver_args.json
for t in edgeconv_f16b conv5_f16_2p knn_mfma chamfer chamfer_c4 emd_sweep emd_match attention; do cp_if gpurun_out/pmc_$t.txt profiles/round5_pmc_$t.txt; done
cp_if gpurun_out/r5_kbench.txt profiles/round5_kbench.txt
cp_if gpurun_out/r5_emd_bench.txt profiles/round5_emd_bench.txt
cp_if gpurun_out/r5_chamfer_bench.txt profiles/round5_chamfer_bench.txt
cp_if gpurun_out/r5_attention_bench.txt profiles/round5_attention_bench.txt
python tools/kernel_meta.py > profiles/round5_kernel_resources.txt 2>/dev/null && echo "  profiles/round5_kernel_resources.txt"
python tools/traffic_json.py 5 > /dev/null && echo "  profiles/round5_traffic.json"
