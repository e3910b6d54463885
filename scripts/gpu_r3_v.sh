#!/bin/bash
# round 3, call V (last): whole GPU suite + smoke + bench at the driver's arguments, then the geometry fuzz, FlowNet3D's kernel table, kbench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
bash scripts/gpu_r3_h.sh 2>&1 | tail -6
timeout 200 python tools/fuzz_geometry.py 3 8 2>&1 | grep -v amdgpu | tail -1
timeout 200 python tools/flownet_profile.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -8
timeout 600 python tools/kbench.py > gpurun_out/r3_kbench.txt 2>&1
grep "three_nn\|knn_pair\|pcn_fwd\|dgcnn_fwd" gpurun_out/r3_kbench.txt | grep -v "^{"
