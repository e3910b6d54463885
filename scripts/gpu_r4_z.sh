#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
bash tools/pmc.sh bq_cells bq_cells_query_kernel > /dev/null 2>&1
cat gpurun_out/pmc_bq_cells.txt
