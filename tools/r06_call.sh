#!/bin/bash
# round 6: is the (delta2, W^T_lo) term of dh needed?  f16x2h = 103 without bit 2 = 99 (and 35 / 67 / 3 for the price list) on the 100-step curves of c2 and the c3 envelope
mkdir -p gpurun_out/r06c49
timeout 900 python tools/curve_modes.py --config c2 --modes f16x2h,f16x2:99,f16x2:67,f16x2:35,f16x2:3 --time > gpurun_out/r06c49/curve_c2.txt 2>&1
timeout 900 python tools/curve_modes.py --config c3 --modes f16x2h,f16x2:99 --time > gpurun_out/r06c49/curve_c3.txt 2>&1
grep -h "^\[\|Error" gpurun_out/r06c49/curve_c2.txt gpurun_out/r06c49/curve_c3.txt | sed 's/; ae max[^;]*;/;/' | cut -c1-330
