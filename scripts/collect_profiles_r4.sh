#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box by scripts/gpu_r4_final.sh) -> profiles/round4_* (tracked)
set -u
R=/root/repo; cd $R
cp_if() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cp_if "$(find gpurun_out/prof_r4 -name '*kernel_stats.csv' | head -1)" profiles/round4_kernel_stats_bench.csv
cp_if "$(find gpurun_out/prof_r4_c5 -name '*kernel_stats.csv' | head -1)" profiles/round4_kernel_stats_bench_c5.csv
cp_if gpurun_out/r4_bench.json profiles/round4_bench.json
cp_if gpurun_out/r4_bench_driver.json profiles/round4_bench_driver_args.json
cp_if gpurun_out/r4_bench_c5.json profiles/round4_bench_c5.json
cp_if gpurun_out/r4_bench_c5_serial.json profiles/round4_bench_c5_serial.json
cp_if gpurun_out/pmc_edgeconv_f16b.txt profiles/round4_pmc_edgeconv_f16b.txt
cp_if gpurun_out/pmc_conv5_f16_2p.txt profiles/round4_pmc_conv5_f16_2p.txt
cp_if gpurun_out/pmc_knn_mfma.txt profiles/round4_pmc_knn_mfma.txt
cp_if gpurun_out/pmc_sa_mlp3.txt profiles/round4_pmc_sa_mlp3.txt
cp_if gpurun_out/r4_kbench.txt profiles/round4_kbench.txt
cp_if gpurun_out/r4_dcp_train_trace.txt profiles/round4_dcp_train_trace.txt
cp_if gpurun_out/r4_bmm_bench.txt profiles/round4_bmm_bench.txt
cp_if gpurun_out/r4_dcp_kernels.txt profiles/round4_dcp_forward_kernels.txt
python tools/kernel_meta.py > profiles/round4_kernel_resources.txt 2>/dev/null && echo "  profiles/round4_kernel_resources.txt"
python tools/traffic_json.py 4 > /dev/null && echo "  profiles/round4_traffic.json"
