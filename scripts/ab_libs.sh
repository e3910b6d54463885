#!/bin/bash
# A/B of whole-library variants inside the bench step on ONE box: scripts/ab_libs.sh <rounds> <name|product> ...   (tools/build_variant_lib.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; rounds=$1; shift
for i in $(seq $rounds); do for v in "$@"; do
  if [ "$v" = product ]; then unset L3D_LIB_PATH; else export L3D_LIB_PATH=$R/tools/bin/libl3d_$v.so; fi
  python $R/bench.py --gpus 1 --steps 100 --warmup 20 --no-other-configs --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-10s %7.0f clouds/s %.4f ms  ec(live) %.1f  b2b: knn %.1f ec - conv5 %.1f ch %.1f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3, k['knn_ms']*1e3, k['conv5_ms']*1e3, k['chamfer_ms']*1e3))"
done; done
