#!/bin/bash
# round 3, call M: the sorted-list Chamfer backward -- bit identity, timings; then the bench line (conv_f16.hip was touched)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_kernels.py tests/test_gpu_grad_routes.py -q -m gpu -p no:cacheprovider -k "chamfer or pcn" 2>&1 | grep -v "^  File\|dist-packages" | tail -15
timeout 300 python tools/cd_bwd_bench.py 2>&1 | tail -6
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'], d['roofline']['frac'])"
