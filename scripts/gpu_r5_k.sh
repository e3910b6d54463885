#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r5k_pytest.log 2>&1; tail -5 gpurun_out/r5k_pytest.log
