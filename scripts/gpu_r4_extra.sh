#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
bash tools/pmc.sh attention attention_f16b_kernel > /dev/null 2>&1
grep "MFMA\|WAVE_CYCLES\|WAIT\|INSTS_VALU\|INSTS_LDS\|GUI_ACTIVE\|FETCH\|WRITE\|BANK" gpurun_out/pmc_attention.txt
timeout 300 python tools/flownet_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-170 > gpurun_out/r4_flownet_profile.txt; head -24 gpurun_out/r4_flownet_profile.txt
