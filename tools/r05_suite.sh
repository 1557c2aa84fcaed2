#!/bin/bash
# the driver's two round-end commands on a fresh box: smoke() and the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r05_smoke.txt; tail -5 gpurun_out/r05_smoke.txt
timeout 2400 python -m pytest tests/ -q -m gpu -x --durations=15 > gpurun_out/r05_gpu_suite.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gpu_suite.txt
tail -40 gpurun_out/r05_gpu_suite.txt
