#!/bin/bash
# round 6: the Gram's three products per K tile in one stage (gram64f_kernel): parity tests, per-kernel A/B, the c2 / c3 curves
mkdir -p gpurun_out/r06c33
O=gpurun_out/r06c33
timeout 900 python -m pytest tests/test_hip_f16.py tests/test_hip_kernels.py -q -x -k "gram or miner or triplet" 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
timeout 300 python tools/kprof.py --precision f16x2h --opt gram_fused=0 --tag walk 2>/dev/null | grep "==\|gram " | cut -c1-140 >> $O/kprof_ab.txt
timeout 300 python tools/kprof.py --precision f16x2h --opt gram_fused=1 --tag fused 2>/dev/null | grep "==\|gram " | cut -c1-140 >> $O/kprof_ab.txt
done
cat $O/kprof_ab.txt
timeout 600 python tools/curve_modes.py --config c2 --modes f16x2h,bf16x3 > $O/curve_c2.txt 2>&1; grep -h "^\[\|Error" $O/curve_c2.txt | sed 's/; ae max[^;]*;/;/'
timeout 600 python tools/curve_modes.py --config c3 --modes f16x2h > $O/curve_c3.txt 2>&1; grep -h "^\[\|Error" $O/curve_c3.txt | sed 's/; ae max[^;]*;/;/'
