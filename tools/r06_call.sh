#!/bin/bash
# round 6: the GPU suite and smoke() at the final HEAD
mkdir -p gpurun_out/r06head
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06head/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06head/gpu_suite.txt
tail -4 gpurun_out/r06head/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06head/smoke.txt 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r06head/smoke.txt
