#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
smi() { rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "junction|Power \(W\)|Socket Power|sclk" | tr '\n' ';' | cut -c1-300; echo; }
echo "idle:"; smi
timeout 900 python -m pytest tests -q -m gpu -x > /dev/null 2>&1; echo "after tests:"; smi
for i in 1 2 3; do
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), round(d['ms_per_step'],4), {a:round(b,4) for a,b in d['kernels'].items() if a.endswith('_ms')})"
  smi
done
