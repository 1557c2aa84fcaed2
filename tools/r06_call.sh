#!/bin/bash
# round 6: the two-term register-carry decode loop (mainloop_n64_c2: f16x2d / f16x2): parity tests, per-kernel A/B on c1's mode, c1 / c5 curves
mkdir -p gpurun_out/r06c29
O=gpurun_out/r06c29
timeout 900 python -m pytest tests/test_hip_f16.py -q -x 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
timeout 300 python tools/kprof.py --precision f16x2d --strategy none --opt decode_x3=0 --tag seg2 2>/dev/null | grep "==\|decode_loss" | cut -c1-140 >> $O/kprof_ab.txt
timeout 300 python tools/kprof.py --precision f16x2d --strategy none --opt decode_x3=1 --tag c2 2>/dev/null | grep "==\|decode_loss" | cut -c1-140 >> $O/kprof_ab.txt
done
timeout 300 python tools/kprof.py --precision f16x2 --opt decode_x3=0 --tag seg2 2>/dev/null | grep "==\|decode_loss" | cut -c1-140 >> $O/kprof_ab.txt
timeout 300 python tools/kprof.py --precision f16x2 --opt decode_x3=1 --tag c2 2>/dev/null | grep "==\|decode_loss" | cut -c1-140 >> $O/kprof_ab.txt
cat $O/kprof_ab.txt
timeout 900 python -m pytest tests/test_hip_long_curves.py tests/test_hip_curves.py -q -x -s 2>&1 | grep -E "curve\]|passed|failed|FAILED|Error|^E " | cut -c1-260 > $O/curves.txt; cat $O/curves.txt
