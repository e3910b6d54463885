#!/bin/bash
# EMD knobs, A/B on one box (rebuilds emd.hip in place): sweep prefetch depth U, match kernel strip LT
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { python -m learning3d_amd.build > /dev/null 2>&1; timeout 300 python tools/emd_bench.py 2>/dev/null | grep -E "emd_fwd_B32_n1024_split[024] |emd_fwd_B64_n1024_split1|BIT|maxdiff" | sed -n '2p;7,10p' | sed "s/^/$1: /"; }
run base
sed -i 's/constexpr int R = 256 \/ S, U = 4,/constexpr int R = 256 \/ S, U = 8,/' learning3d_amd/csrc/emd.hip; run U8
sed -i 's/constexpr int R = 256 \/ S, U = 8,/constexpr int R = 256 \/ S, U = 2,/' learning3d_amd/csrc/emd.hip; run U2
sed -i 's/constexpr int R = 256 \/ S, U = 2,/constexpr int R = 256 \/ S, U = 4,/' learning3d_amd/csrc/emd.hip
sed -i 's/#define EMD_MATCH_LT 64 /#define EMD_MATCH_LT 128/' learning3d_amd/csrc/emd.hip; run LT128
sed -i 's/#define EMD_MATCH_LT 128/#define EMD_MATCH_LT 32 /' learning3d_amd/csrc/emd.hip; run LT32
