#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_fn
cat > /tmp/fn_run.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch
from learning3d_amd.models import FlowNet3D
torch.manual_seed(0)
net = FlowNet3D().cuda().eval()
g = torch.Generator().manual_seed(3)
B, N = 32, 8192
pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2).cuda()
pc2 = (pc1 + 0.05 * torch.randn((B, 3, N), generator=g).cuda()).contiguous()
f1 = torch.rand((B, 3, N), generator=g).cuda(); f2 = torch.rand((B, 3, N), generator=g).cuda()
with torch.no_grad():
    for _ in range(6): net(pc1, pc2, f1, f2)
torch.cuda.synchronize()
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fn -o fn -- python /tmp/fn_run.py > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_fn -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total per forward %.2f ms" % (tot/6/1e6))
for r in rows[:18]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
PY
