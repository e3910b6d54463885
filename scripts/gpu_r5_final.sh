#!/bin/bash
# round 5: the committed evidence set -- bench lines (defaults, driver arguments), rocprofv3 kernel stats of the bench command, PMC passes
# (separate --pmc runs with --kernel-trace only) of the step's kernels and of this round's new kernels, kbench, EMD / Chamfer benches
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > gpurun_out/r5_bench_driver.json 2>> gpurun_out/r5_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5 -o b -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r5.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5_emd -o emd -- python $R/tools/pmc_one.py emd > $R/gpurun_out/prof_r5_emd.log 2>&1 )
bash tools/pmc.sh edgeconv_f16b edgeconv_f16b > /dev/null 2>&1
bash tools/pmc.sh conv5_f16_2p "^(void )?conv_f16_kernel" > /dev/null 2>&1
bash tools/pmc.sh knn_mfma knn_mfma_kernel > /dev/null 2>&1
bash tools/pmc.sh chamfer chamfer_fwd_packed_kernel > /dev/null 2>&1
bash tools/pmc.sh chamfer_c4 chamfer_mfma_kernel > /dev/null 2>&1
bash tools/pmc.sh emd_sweep 'emd_sweep_kernel<2, true>' > /dev/null 2>&1
bash tools/pmc.sh emd_match emd_match_kernel > /dev/null 2>&1
bash tools/pmc.sh attention attention_f16b_kernel > /dev/null 2>&1
timeout 900 python tools/kbench.py > gpurun_out/r5_kbench.txt 2>&1
timeout 300 python tools/emd_bench.py > gpurun_out/r5_emd_bench.txt 2>&1
timeout 300 python tools/chamfer_bench.py > gpurun_out/r5_chamfer_bench.txt 2>&1
timeout 300 python tools/attention_bench.py > gpurun_out/r5_attention_bench.txt 2>&1
for f in $(find gpurun_out/prof_r5 gpurun_out/prof_r5_emd -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-150; done
tail -c 600 gpurun_out/r5_bench.json; echo; head -12 gpurun_out/pmc_edgeconv_f16b.txt; tail -3 gpurun_out/r5_bench.err
