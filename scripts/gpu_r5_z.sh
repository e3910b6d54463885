#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('bench', round(j['value']), round(j['ms_per_step'],4), 'frac', round(r['frac'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in r['sustained'].items() if k!='note'})"; done
tools/bin/pclk > gpurun_out/r5_mfma_clock.txt; cat gpurun_out/r5_mfma_clock.txt | head -8
timeout 300 python -m pytest tests -m gpu -x -q -k "integration or native" 2>&1 | tail -2
