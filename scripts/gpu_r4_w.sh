#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-250 > gpurun_out/r4_pytest_w.txt
cat gpurun_out/r4_pytest_w.txt
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 > gpurun_out/r4_bench_c5.json 2>gpurun_out/r4_bench_c5.err
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 --c5-serial --no-cpu-baseline > gpurun_out/r4_bench_c5_serial.json 2>>gpurun_out/r4_bench_c5.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4_c5 -o c5 -- python $R/bench.py --workload c5 --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_r4_c5.log 2>&1 )
bash tools/pmc.sh sa_mlp3 sa_mlp3_kernel > /dev/null 2>&1
python - <<P
import json
for f in ("gpurun_out/r4_bench_c5.json","gpurun_out/r4_bench_c5_serial.json"):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, j["value"], j["ms_per_step"], j["serial_ms_per_step"], j["kernels"], j["roofline"]["frac"])
P
head -6 $(find gpurun_out/prof_r4_c5 -name "*kernel_stats.csv" | head -1) | cut -c1-170
grep "BANK_CONFLICT\|LDS_IDX_ACTIVE\|FETCH_SIZE\|WRITE_SIZE" gpurun_out/pmc_sa_mlp3.txt
