#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_kernels.py tests/test_gpu_config_size.py tests/test_gpu_integration_shims.py -m gpu -q -x -k "ball or set_abstraction or flownet or c5 or shim or config5 or fps" 2>&1 | tail -25 | cut -c1-250 > gpurun_out/r4_pytest_u.txt
cat gpurun_out/r4_pytest_u.txt
for d in 2 3; do
timeout 200 python bench.py --workload c5 --steps 100 --warmup 10 --c5-depth $d --no-cpu-baseline > gpurun_out/r4_bench_c5_u$d.json 2>gpurun_out/r4_bench_c5_u.err
python - <<P
import json
j=json.loads(open("gpurun_out/r4_bench_c5_u$d.json").read().strip().splitlines()[-1])
print("depth $d", j["value"], j["ms_per_step"], j["serial_ms_per_step"], j["kernels"], j["roofline"]["frac"], j["digest"])
P
done
tail -3 gpurun_out/r4_bench_c5_u.err | grep -v amdgpu.ids; true
