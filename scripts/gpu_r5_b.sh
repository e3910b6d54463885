#!/bin/bash
# round 5, call b: EMD backward (bit-exact kernels), per-kernel trace of the EMD forward, PMC of the sweeps
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python tools/emd_bench.py > gpurun_out/r5b_emd_bench.txt 2>&1; cat gpurun_out/r5b_emd_bench.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5_emd
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5_emd -o emd -- python $R/tools/pmc_one.py emd > $R/gpurun_out/r5b_prof.log 2>&1
f=$(find $R/gpurun_out/prof_r5_emd -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -i "emd\|Name" "$f" | cut -c1-220
cd $R
PMC_SETS="1 2" bash tools/pmc.sh emd_sweep 'emd_sweep_kernel<2, true>' > /dev/null 2>&1; cat gpurun_out/pmc_emd_sweep.txt
