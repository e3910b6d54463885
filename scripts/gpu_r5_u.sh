#!/bin/bash
# A/B on one box: conv5's output stores nontemporal or not -- what the NEXT kernels of the step (kNN, EdgeConv) pay for conv5's 134 MB of dirty lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
b() { timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(j['value']), round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['kernels'].items() if k.endswith('_ms')})"; }
p() { timeout 300 python tools/ec_instep_probe.py 2>/dev/null | tail -6 | sed "s/^/$1: /"; }
b base; b base; p base
sed -i 's/^#ifdef CF_NT_STORE/#if 1/' learning3d_amd/csrc/conv_f16.hip; python -m learning3d_amd.build > /dev/null 2>&1
b nt; b nt; p nt
sed -i 's/^#if 1$/#ifdef CF_NT_STORE/' learning3d_amd/csrc/conv_f16.hip; python -m learning3d_amd.build > /dev/null 2>&1
b base; b base
