#!/bin/bash
# round 4: the committed evidence set -- bench lines (defaults, driver arguments, --workload c5 pipelined and serial), rocprofv3 kernel
# stats of the bench commands, PMC passes (separate --pmc runs with --kernel-trace only) of the step's kernels, kbench, DCP training trace
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver.json 2>> gpurun_out/r4_bench.err
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 > gpurun_out/r4_bench_c5.json 2>> gpurun_out/r4_bench.err
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 --c5-serial --no-cpu-baseline > gpurun_out/r4_bench_c5_serial.json 2>> gpurun_out/r4_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4 -o b -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r4.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4_c5 -o c5 -- python $R/bench.py --workload c5 --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_r4_c5.log 2>&1 )
bash tools/pmc.sh edgeconv_f16b edgeconv_f16b > /dev/null 2>&1
bash tools/pmc.sh conv5_f16_2p "^(void )?conv_f16_kernel" > /dev/null 2>&1
bash tools/pmc.sh knn_mfma knn_mfma_kernel > /dev/null 2>&1
bash tools/pmc.sh sa_mlp3 sa_mlp3_kernel > /dev/null 2>&1
timeout 900 python tools/kbench.py > gpurun_out/r4_kbench.txt 2>&1
timeout 300 python tools/train_step_profile.py --only dcp > gpurun_out/r4_dcp_train_trace.txt 2>&1
for f in $(find gpurun_out/prof_r4 gpurun_out/prof_r4_c5 -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-150; done
tail -c 900 gpurun_out/r4_bench.json; echo; head -30 gpurun_out/pmc_edgeconv_f16b.txt; tail -3 gpurun_out/r4_bench.err
