#!/bin/bash
# A/B on one box: knn_mfma with / without s_setprio 3 (rebuilds the one object in place on the box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
b() { timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(j['value']), round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['kernels'].items() if k.endswith('_ms')})"; }
b prio3; b prio3
sed -i 's/#ifndef KM_NO_SETPRIO/#if 0/' learning3d_amd/csrc/knn_mfma.hip; python -m learning3d_amd.build > /dev/null 2>&1
b noprio; b noprio
sed -i 's/#if 0/#ifndef KM_NO_SETPRIO/' learning3d_amd/csrc/knn_mfma.hip; python -m learning3d_amd.build > /dev/null 2>&1
b prio3; b prio3
