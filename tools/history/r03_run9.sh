#!/bin/bash
set -u
O=gpurun_out/r3q
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_step.py -q -x -k "dw_ or fused_optimizer or step_optimizers or full_size" > $O/tests_dw.log 2>&1
tail -6 $O/tests_dw.log | cut -c1-300
for o in "" "--glds -8" "" "--glds -8"; do
  timeout 200 python tools/kprof.py $o >> $O/kprof.txt 2>&1
done
grep -v amdgpu.ids $O/kprof.txt | grep "==\|dw_gemm"
