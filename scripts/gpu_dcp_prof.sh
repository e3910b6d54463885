#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_dcp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dcp -o dcp -- python $R/tools/dcp_breakdown.py > $R/gpurun_out/dcp_prof.log 2>&1
f=$(find $R/gpurun_out/prof_dcp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
PY
