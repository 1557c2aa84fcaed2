#!/bin/bash
# round 6, GPU call 10: which lo terms does c1 (strategy none) need to hold 1e-4 over 100 steps?  and f16x2h candidates on c2
mkdir -p gpurun_out/r06c10
O=gpurun_out/r06c10
timeout 900 python tools/curve_modes.py --config c1 --modes f16x2:13,f16x2:37,f16x2:133,f16x2:45,f16x2:141,f16x2:165,f16x2:173,f16x3 --time > $O/curve_c1_masks.txt 2>&1
timeout 900 python tools/curve_modes.py --config c2 --modes f16x2:69,f16x2:21,f16x2:85,f16x2:375,f16x2:471 --time > $O/curve_c2_masks.txt 2>&1
grep -h "^\[" $O/curve_c1_masks.txt $O/curve_c2_masks.txt
