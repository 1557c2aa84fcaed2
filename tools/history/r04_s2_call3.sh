#!/bin/bash
# round-4 (second session) call 3: all product terms restored; the split-mode step / config / data-parallel tests at small shapes with the
# paired dW stages (default) and without, the gradient-only form of the split dW kernel (phase 1), per-kernel timings of the final default
O=gpurun_out/${1:-r4g}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T0=$(date +%s)
timeout 400 python3 -m pytest tests/test_hip_step.py tests/test_hip_configs.py tests/test_hip_dp.py tests/test_hip_full_curve.py -q -m gpu -p no:cacheprovider -s \
    -k "x3" > $O/pytest_x3.log 2>&1; echo "pytest_x3 rc=$? t=$(( $(date +%s) - T0 ))" > $O/rc.txt
grep -a "\[curve\]\|passed\|failed\|FAILED\|Error" $O/pytest_x3.log | cut -c1-260 | tail -40 > $O/pytest_x3_tail.txt
K="timeout 120 python3 tools/kprof.py --precision bf16x3"
$K --tag default > $O/kprof_default.txt 2>&1
$K --opt dw_pair=0 --tag unpaired > $O/kprof_unpaired.txt 2>&1
$K --phase 1 --tag phase1 > $O/kprof_phase1.txt 2>&1
$K --phase 1 --opt fused_opt=0 --tag phase1_unfused > $O/kprof_phase1_unfused.txt 2>&1
echo "kprof done t=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt; cat $O/pytest_x3_tail.txt | tail -25
grep -a "==" -A9 $O/kprof_default.txt | grep -v amdgpu | cut -c1-150
grep -a "==\|dw_gemm\|opt_step" $O/kprof_unpaired.txt $O/kprof_phase1.txt $O/kprof_phase1_unfused.txt | cut -c1-170
