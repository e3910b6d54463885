#!/bin/bash
# usage: tools/pmc.sh <kernel-tag> <kernel-name-regex>   -> gpurun_out/pmc_<tag>_*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; RE=$2
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE GRBM_COUNT" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "${PMC_SETS:-}" ] && ! echo " $PMC_SETS " | grep -q " $i "; then continue; fi      # PMC_SETS="4 5": only those counter passes
  rm -rf $R/gpurun_out/pmc_tmp
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_tmp -o p -- python $R/tools/pmc_one.py $TAG > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" "$RE" <<'PY' > $R/gpurun_out/pmc_${TAG}_$i.txt
import csv, sys, re, collections
f, rx = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if rx.search(r["Kernel_Name"]):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
  fi
done
rm -rf $R/gpurun_out/pmc_tmp
cat $R/gpurun_out/pmc_${TAG}_*.txt > $R/gpurun_out/pmc_${TAG}.txt; cat $R/gpurun_out/pmc_${TAG}.txt
