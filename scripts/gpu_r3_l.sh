#!/bin/bash
# round 3, call L: the step's Chamfer branch on a second stream -- where should it leave the chain?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for rep in 1 2; do
for f in none start edgeconv conv5; do
  timeout 300 python bench.py --no-cpu-baseline --fork $f > gpurun_out/r3l_$f.$rep.json 2> gpurun_out/r3l_$f.$rep.err
  echo "fork=$f rep=$rep $(tail -1 gpurun_out/r3l_$f.$rep.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels'), d['roofline']['frac'])" 2>&1 | tail -1)"
  tail -3 gpurun_out/r3l_$f.$rep.err
done
done
for f in none conv5; do
  timeout 300 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 --fork $f > gpurun_out/r3l_drv_$f.json 2>> gpurun_out/r3l_drv.err
  echo "driver-args fork=$f $(tail -1 gpurun_out/r3l_drv_$f.json | cut -c1-220)"
done
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r3l_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3l_tests.log
grep -v "^  File\|dist-packages" gpurun_out/r3l_tests.log | tail -15
