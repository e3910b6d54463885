#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { python -m learning3d_amd.build > /dev/null 2>&1; timeout 300 python tools/emd_bench.py 2>/dev/null | grep -E "emd_fwd_B32_n1024_split[02] |emd_fwd_B64_n1024_split1|emd_bwd_B32" | sed "s/^/$1: /"; }
run base
sed -i 's/if (pair_store) \*(float2 \*)o = make_float2(acc.x, acc.y);/if (pair_store) __builtin_nontemporal_store(acc, (f32x2 *)o);/' learning3d_amd/csrc/emd.hip; run nt
run nt
