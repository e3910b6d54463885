#!/bin/bash
# round 4, after --fork knn became the default: the bench lines and the kernel stats of the bench command again (the PMC passes, kbench
# and the c5 lines of scripts/gpu_r4_final.sh are unaffected), plus the N > 1 code path over RCCL with one rank
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_rccl_one_rank.py tests/test_gpu_two_ranks.py -m gpu -q 2>&1 | tail -2 | cut -c1-200
timeout 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver.json 2>> gpurun_out/r4_bench.err
timeout 300 python bench.py --fork none --no-cpu-baseline --no-other-configs > gpurun_out/r4_bench_fork_none.json 2>> gpurun_out/r4_bench.err
rm -rf gpurun_out/prof_r4
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4 -o b -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r4.log 2>&1 )
for f in gpurun_out/r4_bench.json gpurun_out/r4_bench_driver.json gpurun_out/r4_bench_fork_none.json; do python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(j['value']), round(j['ms_per_step'],4), round(j['roofline']['frac'],3), j['config'].get('chamfer_branch','')[:40], j['kernels']['edgeconv_ms'])"; done
head -6 $(find gpurun_out/prof_r4 -name "*kernel_stats.csv" | head -1) | cut -c1-150
