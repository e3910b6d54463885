#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "chamfer or Chamfer or c4 or pcn" 2>&1 | tail -2 | cut -c1-200
for rep in 1 2; do for f in none knn; do
  timeout 300 python bench.py --fork $f --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fork $f', round(j['value']), round(j['ms_per_step'],4), j['loss'], round(j['kernels']['chamfer_ms']*1e3,1))"
done; done
