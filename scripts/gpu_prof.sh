#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; prints the top of the stats table
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline ${2:-} > $R/gpurun_out/rocprof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s}%")
PY
tail -1 $R/gpurun_out/rocprof_$TAG.log | cut -c1-160
