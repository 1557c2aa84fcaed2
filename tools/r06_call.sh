#!/bin/bash
# round 6, GPU call 12: cheaper lo-term masks for batch_all over 100 steps (all contain the two W terms)
mkdir -p gpurun_out/r06c12
O=gpurun_out/r06c12
timeout 900 python tools/curve_modes.py --config c2 --modes f16x2:7,f16x2:39,f16x2:23,f16x2:71,f16x2:103,f16x2:55,f16x2:119,f16x2:141,f16x2:143 --time > $O/curve_c2_masks2.txt 2>&1
grep -h "^\[" $O/curve_c2_masks2.txt | sed 's/; ae max[^;]*;/;/'
