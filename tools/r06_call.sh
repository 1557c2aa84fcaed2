#!/bin/bash
# round 6: A/B on ONE box -- the library of the session's first commit (9ebc342, before DAE_LAUNCH) against HEAD's, un-profiled c2 step, alternating
mkdir -p gpurun_out/r06c50
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then L="--lib-f16 tools/ab_old/libdae_hip_f16_old.so"; else L=""; fi
    timeout 200 python tools/kprof.py --precision f16x2h $L --tag $which 2>/dev/null | grep "^==" | sed 's/info=.*//' >> gpurun_out/r06c50/ab.txt
  done
done
cat gpurun_out/r06c50/ab.txt
