#!/bin/bash
# kernel profiles of the remaining model forwards: anything that is not an l3d kernel in an inference path is a leftover
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
cat > /tmp/misc_run.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch
from learning3d_amd.models import PointNet, Classifier, DGCNN
from learning3d_amd.models.prnet import DGCNN as PRDGCNN
which = sys.argv[1]
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda()
if which == "pointnet":
    net = Classifier(feature_model=PointNet(emb_dims=1024, use_bn=True)).cuda().eval()
elif which == "prnet":
    net = PRDGCNN(emb_dims=512).cuda().eval(); x = x.permute(0, 2, 1).contiguous()
with torch.no_grad():
    for _ in range(8): net(x)
torch.cuda.synchronize()
PY
for W in pointnet prnet; do
  rm -rf $R/gpurun_out/prof_$W
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o $W -- python /tmp/misc_run.py $W > $R/gpurun_out/prof_$W.log 2>&1
  f=$(find $R/gpurun_out/prof_$W -name "*kernel_stats.csv" | head -1)
  echo "== $W"; tail -2 $R/gpurun_out/prof_$W.log | cut -c1-200
  python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total per forward %.3f ms" % (tot/8/1e6))
for r in rows[:10]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
PY
done
