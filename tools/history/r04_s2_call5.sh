#!/bin/bash
# round-4 (second session) call 5: gemm_nt_pc with the LDS-staged 16-byte slab epilogue -- GEMM numerics tests, the step tests that run on it
# (Gram, dh, dense encode), per-kernel timings in both modes
O=gpurun_out/${1:-r4j}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T0=$(date +%s)
timeout 300 python3 -m pytest tests/test_hip_kernels.py tests/test_hip_step.py tests/test_hip_configs.py -q -m gpu -p no:cacheprovider \
    -k "gemm or x3 or config or bf16_matches or fp32_matches" > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - T0 ))" > $O/rc.txt
grep -a "passed\|failed\|FAILED\|Error" $O/pytest.log | cut -c1-200 | tail -20 > $O/pytest_tail.txt
timeout 120 python3 tools/kprof.py --precision bf16x3 --tag x3 > $O/kprof_x3.txt 2>&1
timeout 120 python3 tools/kprof.py --precision bf16 --tag bf16 > $O/kprof_bf16.txt 2>&1
timeout 120 python3 tools/kprof.py --precision fp32 --tag fp32 > $O/kprof_fp32.txt 2>&1
echo "done t=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt $O/pytest_tail.txt; grep -a -v amdgpu $O/kprof_x3.txt $O/kprof_bf16.txt | cut -c1-150; grep -a "==" $O/kprof_fp32.txt | cut -c1-150
