#!/bin/bash
# round 3, call B: failing tests in isolation with full logs + gradient diagnostics
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_grad_routes.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r3b_routes.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -s -k "lpfa" > gpurun_out/r3b_lpfa.log 2>&1
timeout 300 python tools/grad_diag.py > gpurun_out/r3b_diag.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_curvenet_lpfa_golden --ignore tests/test_gpu_grad_routes.py > gpurun_out/r3b_rest.log 2>&1
grep -v "^  File\|site-packages\|dist-packages" gpurun_out/r3b_routes.log | tail -150; tail -30 gpurun_out/r3b_lpfa.log; cat gpurun_out/r3b_diag.log; tail -40 gpurun_out/r3b_rest.log
