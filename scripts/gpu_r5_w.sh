#!/bin/bash
# A/B on one box: EdgeConv's plane image stores nontemporal or not (conv5 reads that image right behind it)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
b() { timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(j['value']), round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['kernels'].items() if k.endswith('_ms')})"; }
b base; b base
sed -i 's/^#ifdef EF_NT_OUT /#if 1 \/\/EF_NT_OUT /' learning3d_amd/csrc/edgeconv_f16b.hip; python -m learning3d_amd.build > /dev/null 2>&1
b ecnt; b ecnt
timeout 200 python tools/ec_instep_probe.py 2>/dev/null | tail -6 | sed "s/^/ecnt: /"
sed -i 's/^#if 1 \/\/EF_NT_OUT /#ifdef EF_NT_OUT /' learning3d_amd/csrc/edgeconv_f16b.hip; python -m learning3d_amd.build > /dev/null 2>&1
b base; b base
