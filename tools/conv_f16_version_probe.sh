#!/bin/bash
# Stand-alone timing of conv_f16.hip AS IT WAS at given commits (plus the working tree): each version is compiled by hipcc
# into its own binary that launches the kernel at PCN's conv4 shape (B = 64, N = 2048, 512 -> 1024) with fp32 output and with the
# 128-point pool.  How the regression of LABLOG R2.4h was bisected.
#   usage (here):      tools/conv_f16_version_probe.sh build d1b5533 07f305d 0baea48      -> tools/bin/cfv/probe_<hash|cur>
#   usage (GPU box):   tools/conv_f16_version_probe.sh run
set -u
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/tools/bin/cfv
if [ "${1:-}" = "run" ]; then for p in $D/probe_*; do [ -x "$p" ] && echo "$(basename $p)" && timeout 60 "$p"; done; exit 0; fi
shift; mkdir -p $D
cat > $D/probe_main.inc <<'C'
#include <cstdio>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 64, N = 2048, Cin = 512, Cout = 1024;
    const size_t xb = l3d_f16_act_bytes((long)B * N, Cin), wb = l3d_conv_f16_weight_bytes(Cout, Cin);
    void *x, *w; float *y, *yp;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&y, (size_t)B * Cout * N * 4); hipMalloc(&yp, (size_t)B * Cout * (N / 8) * 4);
    hipMemset(x, 0x11, xb); hipMemset(w, 0x11, wb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            for (int it = 0; it < 3; it++) { if (mode) POOLCALL; else l3d_pointwise_conv_f16(x, w, nullptr, nullptr, 0, B, Cin, Cout, N, 1, y, nullptr); }
            hipEventRecord(e0, nullptr);
            for (int it = 0; it < 10; it++) { if (mode) POOLCALL; else l3d_pointwise_conv_f16(x, w, nullptr, nullptr, 0, B, Cin, Cout, N, 1, y, nullptr); }
            hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("  %s: %.1f us\n", mode ? "pool-only" : "fp32 out ", best * 100.f);
    }
    return 0;
}
C
for h in "$@" cur; do
  mkdir -p $D/$h/x/csrc $D/$h/include
  if [ $h = cur ]; then cp $R/learning3d_amd/csrc/conv_f16.hip $D/$h/x/csrc/; cp $R/include/l3d_hip.h $D/$h/include/
  else git -C $R show $h:learning3d_amd/csrc/conv_f16.hip > $D/$h/x/csrc/conv_f16.hip; git -C $R show $h:include/l3d_hip.h > $D/$h/include/l3d_hip.h; fi
  cp $R/learning3d_amd/csrc/common.h $R/learning3d_amd/csrc/split_bf16.h $R/learning3d_amd/csrc/split_f16.h $D/$h/x/csrc/
  # the pool entry point gained its run-length argument with the template version
  if grep -q "float \*ypool, int pool, l3d_stream_t" $D/$h/include/l3d_hip.h; then PA=", 128"; else PA=""; fi
  printf '#include "x/csrc/conv_f16.hip"\n#define POOLCALL l3d_pointwise_conv_f16_pool(x, w, nullptr, nullptr, 0, nullptr, B, Cin, Cout, N, 1, nullptr, yp%s, nullptr)\n#include "../probe_main.inc"\n' "$PA" > $D/$h/probe.hip
  (cd $D/$h && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off probe.hip -o ../probe_$h 2>&1 | grep -E " error" | head -3)
done
ls $D | grep probe_ | grep -v inc
