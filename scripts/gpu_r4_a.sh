#!/bin/bash
# round 4, call A: two-waves-per-SIMD MFMA/VALU probe, EdgeConv variants (round-3 kernel vs the staged-output kernel), GPU suite, PMC of the new kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 120 tools/bin/p2w > gpurun_out/r4_p2w.txt 2>&1
timeout 300 python tools/variant_lab.py run ef r3 new nofin > gpurun_out/r4_lab_ef_a.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_pytest_a.txt
bash tools/pmc.sh edgeconv_f16b edgeconv_f16b > /dev/null 2>&1
cat gpurun_out/r4_p2w.txt; cat gpurun_out/r4_lab_ef_a.txt; tail -5 gpurun_out/r4_pytest_a.txt; cat gpurun_out/pmc_edgeconv_f16b.txt
