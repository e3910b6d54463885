#!/bin/bash
# One parameterised GPU-box script (replaces the per-experiment scripts/gpu_r*_*.sh of earlier rounds):
#     gpurun -- 'bash scripts/gpu.sh <task> [<task> ...]'      every task writes under gpurun_out/
# tasks: tests | tests:<pytest -k expr> | bench | bench_driver | timeline | stats | pmc:<tag>:<kernel regex> | kbench |
#        tool:<tools/NAME.py args,comma,separated> | lab:<family>:<variants,comma> | sh:<command>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for task in "$@"; do
  name=${task%%:*}; arg=${task#*:}; [ "$arg" = "$task" ] && arg=""
  echo "=== $task"
  case $name in
    tests)  if [ -n "$arg" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$arg" 2>&1 | tail -15;
            else timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15; fi ;;
    bench)  timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json | head -c 1500; echo; tail -3 gpurun_out/bench.err ;;
    bench_driver) for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/bench_driver_$i.json 2>> gpurun_out/bench.err;
              python - gpurun_out/bench_driver_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{d['value']:.0f} clouds/s  {d['ms_per_step']:.4f} ms  frac {r['frac']:.3f}  sustained {r.get('sustained', {}).get('frac_of_sustained')}  "
      f"{r.get('sustained', {}).get('shader_mhz')} MHz  kernels {d['kernels']['knn_ms']:.4f} {d['kernels']['edgeconv_ms']:.4f} {d['kernels']['conv5_ms']:.4f}")
PY
            done ;;
    timeline) ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/tl && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o t -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $R/gpurun_out/tl.log 2>&1 )
            python tools/step_timeline.py "$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)" | tee gpurun_out/step_timeline.txt ;;
    stats)  ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o b -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof.log 2>&1 )
            f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/kernel_stats_bench.csv; head -8 "$f" | cut -c1-160 ;;
    pmc)    tag=${arg%%:*}; re=${arg#*:}; bash tools/pmc.sh "$tag" "$re" > /dev/null 2>&1; head -40 gpurun_out/pmc_$tag.txt ;;
    kbench) timeout 900 python tools/kbench.py > gpurun_out/kbench.txt 2>&1; tail -40 gpurun_out/kbench.txt ;;
    tool)   prog=${arg%%,*}; rest=${arg#*,}; [ "$rest" = "$arg" ] && rest=""; out=gpurun_out/$(basename $prog .py).txt
            timeout 900 python tools/$prog ${rest//,/ } > $out 2>&1; tail -40 $out ;;
    lab)    fam=${arg%%:*}; vars=${arg#*:}; timeout 900 python tools/variant_lab.py run $fam ${vars//,/ } 2>&1 | tee gpurun_out/lab_$fam.txt | tail -30 ;;
    sh)     timeout 1200 bash -c "$arg" 2>&1 | tail -40 ;;
    *)      echo "unknown task $task" ;;
  esac
done
