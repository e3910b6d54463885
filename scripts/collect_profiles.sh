#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box by scripts/gpu.sh tasks) -> profiles/round<N>_* (tracked).  usage: collect_profiles.sh <N>
set -u
R=/root/repo; cd $R; N=${1:?round number}
cp_if() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cp_if gpurun_out/kernel_stats_bench.csv profiles/round${N}_kernel_stats_bench.csv
cp_if gpurun_out/bench.json profiles/round${N}_bench.json
cp_if gpurun_out/bench_driver_1.json profiles/round${N}_bench_driver_args.json
cp_if gpurun_out/step_timeline.txt profiles/round${N}_step_timeline.txt
for f in gpurun_out/pmc_*.txt; do t=$(basename $f .txt); case $t in pmc_*_[0-9]) ;; *) cp_if $f profiles/round${N}_$t.txt ;; esac; done
for t in kbench featknn_bench dcp_kernels flownet_bench scatter_det_bench emd_bench attention_bench chamfer_bench fk_timeline; do cp_if gpurun_out/$t.txt profiles/round${N}_$t.txt; done
python tools/kernel_meta.py > profiles/round${N}_kernel_resources.txt 2>/dev/null && echo "  profiles/round${N}_kernel_resources.txt"
python tools/traffic_json.py $N > /dev/null && echo "  profiles/round${N}_traffic.json"
