#!/bin/bash
# round 3, call W: PMC passes (separate --pmc runs, kernel trace only) of the two round-3 kNN kernels at config 5's shapes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out; : > $R/gpurun_out/pmc_knn_c5.txt
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_tmp
  timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_tmp -o p -- python $R/tools/pmc_knn_c5.py > $R/gpurun_out/pmc_knn_c5_$i.log 2>&1
  f=$(find /tmp/pmc_tmp -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' >> $R/gpurun_out/pmc_knn_c5.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for tag in ("knn_select_kernel", "knn_small_kernel"):
        if tag in n:
            acc[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (tag, k), v in sorted(acc.items()):
    print(f"{tag:20s} {k:24s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
done
cat $R/gpurun_out/pmc_knn_c5.txt
