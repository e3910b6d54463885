#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r2d_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2d_tests.log
timeout 300 python tools/knn_bench.py > gpurun_out/knn_bench.log 2>&1; cat gpurun_out/knn_bench.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2d_bench.log 2>&1; tail -1 gpurun_out/r2d_bench.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_knn
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_knn -o knn -- python $R/tools/knn_bench.py > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_knn -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
