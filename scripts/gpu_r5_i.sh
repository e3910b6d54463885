#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
PMC_SETS="1 2 3" bash tools/pmc.sh chamfer_c4 chamfer_mfma_kernel > /dev/null 2>&1; cat gpurun_out/pmc_chamfer_c4.txt
