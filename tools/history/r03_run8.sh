#!/bin/bash
set -u
O=gpurun_out/r3o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dp.py -q -x > $O/tests_dp.log 2>&1
tail -5 $O/tests_dp.log | cut -c1-300
timeout 200 python tools/dp_step_breakdown.py > $O/dp_breakdown_fp32.txt 2>&1; grep -v amdgpu.ids $O/dp_breakdown_fp32.txt
timeout 200 python tools/dp_step_breakdown.py --grad-dtype bf16 > $O/dp_breakdown_bf16.txt 2>&1; grep -v amdgpu.ids $O/dp_breakdown_bf16.txt
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "gemm_nt_256" > $O/tests_gemm.log 2>&1; tail -2 $O/tests_gemm.log
