#!/bin/bash
# round 3: the committed evidence set -- bench lines (defaults, driver arguments, --workload c5), rocprofv3 kernel stats of the
# bench command, PMC passes (separate --pmc runs with --kernel-trace only) of the step's kernels, kbench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_driver.json 2>> gpurun_out/r3_bench.err
timeout 300 python bench.py --workload c5 --steps 50 --warmup 10 > gpurun_out/r3_bench_c5.json 2>> gpurun_out/r3_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3 -o b -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_r3.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_c5 -o c5 -- python $R/bench.py --workload c5 --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_r3_c5.log 2>&1 )
bash tools/pmc.sh edgeconv_f16b edgeconv_f16b > /dev/null 2>&1
bash tools/pmc.sh conv5_f16_2p "^(void )?conv_f16_kernel" > /dev/null 2>&1
bash tools/pmc.sh knn_mfma knn_mfma_kernel > /dev/null 2>&1
bash tools/pmc.sh group_c5 group_concat_kernel > /dev/null 2>&1
timeout 900 python tools/kbench.py > gpurun_out/r3_kbench.txt 2>&1
for f in $(find gpurun_out/prof_r3 gpurun_out/prof_r3_c5 -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-150; done
tail -c 600 gpurun_out/r3_bench.json; echo; cat gpurun_out/pmc_edgeconv_f16b.txt | head -40
