#!/bin/bash
# round 3, call U: rocprofv3 kernel stats of tools/kbench.py (every secondary kernel's duration in one table)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_kbench -o k -- python $R/tools/kbench.py > $R/gpurun_out/prof_r3_kbench.log 2>&1 )
f=$(find gpurun_out/prof_r3_kbench -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{float(r['AverageNs'])/1e3:10.1f} us x{r['Calls']:>5s}  {r['Name'][:110]}")
PY
# keep only the stats table (the trace itself is large)
find gpurun_out/prof_r3_kbench -type f ! -name "*kernel_stats.csv" -delete
