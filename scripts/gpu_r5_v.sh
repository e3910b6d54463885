#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "dgcnn or conv or config_size or pointwise or f16" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', round(j['value']), round(j['ms_per_step'],4), round(j['roofline']['frac'],3), {k:round(v,4) for k,v in j['kernels'].items() if k.endswith('_ms')})"; done
timeout 300 python tools/variant_lab.py run cf base 2>&1 | tail -4
