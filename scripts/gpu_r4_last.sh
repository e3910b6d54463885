#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'], j['cpu_baseline']['kind'], list(j['other_configs'].keys()))"
