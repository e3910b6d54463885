#!/bin/bash
# round 4, call D: absolute MFMA rate by waves per SIMD; which instruction-fetch counters this rocprofv3 offers; PMC of the W8 kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 120 tools/bin/pabs > gpurun_out/r4_pabs.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --list-avail 2>&1 | grep -i -E "ifetch|icache|INST_CACHE|SQC_|SQ_WAIT|SQ_INST_LEVEL|SQ_BUSY|SQ_ACTIVE" | head -80 ) > gpurun_out/r4_counters.txt 2>&1
bash tools/pmc.sh edgeconv_f16b edgeconv_f16b > /dev/null 2>&1
cat gpurun_out/r4_pabs.txt; cat gpurun_out/r4_counters.txt | cut -c1-200; cat gpurun_out/pmc_edgeconv_f16b.txt
