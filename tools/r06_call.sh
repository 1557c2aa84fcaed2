#!/bin/bash
# round 6, GPU call 4: gemm_decode_ast without scratch spills -- parity, then timing probes
mkdir -p gpurun_out/r06c4
O=gpurun_out/r06c4
timeout 300 python -m pytest tests/test_hip_kernels.py -k "decode" -q -x > $O/tests_decode.txt 2>&1; echo "rc $?" >> $O/tests_decode.txt
tail -3 $O/tests_decode.txt
for d in 0 1 8 16 24 25 27 31; do
  timeout 200 python tools/kprof.py --precision f16x2 --strategy none --tag "dbg=$d" --glds $((-500000 - d)) 2>&1 | grep -E "^==|decode_loss" | sed 's/info=.*//' >> $O/probe.txt
done
timeout 200 python tools/kprof.py --precision f16x2 --opt decode_ast=0 2>&1 | grep -E "^==|decode_loss" | sed 's/info=.*//' >> $O/probe.txt
timeout 200 python tools/kprof.py --precision f16x2 2>&1 | grep -E "^==|decode_loss" | sed 's/info=.*//' >> $O/probe.txt
timeout 200 python tools/kprof.py --precision bf16 2>&1 | grep -E "^==|decode_loss" | sed 's/info=.*//' >> $O/probe.txt
cat $O/probe.txt
