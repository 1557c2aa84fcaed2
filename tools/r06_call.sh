#!/bin/bash
# round 6: A/B probe -- de-phased start of the decode's first wave of workgroups (dae_set_glds(-500000 - bits)); per-kernel times by dispatch stamps
mkdir -p gpurun_out/r06c48
for bits in 0 10 20 35 74 84 99; do
  timeout 200 python tools/kprof.py --stamps --precision f16x2h --glds $((-500000 - bits)) --tag stagger$bits 2>/dev/null | grep -E "^==|decode_loss|dh_gemm" | sed 's/info=.*//' >> gpurun_out/r06c48/stagger.txt
done
cat gpurun_out/r06c48/stagger.txt
