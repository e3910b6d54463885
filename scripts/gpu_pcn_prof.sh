#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pcn
cat > /tmp/pcn_run.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch
from learning3d_amd.models import PCN
g = torch.Generator().manual_seed(0)
pcn = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).cuda().eval()
part = (torch.rand((64, 2048, 3), generator=g) - 0.5).cuda()
with torch.no_grad():
    for _ in range(12): pcn(part)
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pcn -o pcn -- python /tmp/pcn_run.py > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_pcn -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
PY
