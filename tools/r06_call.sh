#!/bin/bash
# round 6: s_setprio(2) around the MFMA groups of the x3 decode loop -- A/B against HEAD is the previous call's numbers; here: kprof x 3 with the seg walk as the same-box yardstick
mkdir -p gpurun_out/r06c42
O=gpurun_out/r06c42
for i in 1 2 3; do
timeout 300 python tools/kprof.py --precision f16x2h --opt decode_x3=0 --tag seg3 2>/dev/null | grep "==\|decode_loss" | cut -c1-130 >> $O/kprof_ab.txt
timeout 300 python tools/kprof.py --precision f16x2h --opt decode_x3=1 --tag x3prio 2>/dev/null | grep "==\|decode_loss" | cut -c1-130 >> $O/kprof_ab.txt
done
cat $O/kprof_ab.txt
