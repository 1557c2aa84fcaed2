#!/bin/bash
# round 6: the sparse encode from the 16-bit hi image of W alone (encode_w32 = 0) in the split modes -- half the W-row bytes; does the curve hold?
mkdir -p gpurun_out/r06c18
O=gpurun_out/r06c18
timeout 600 python tools/curve_modes.py --config c2 --modes f16x2h,f16x2h::encode_w32=0 --time > $O/enc16_c2.txt 2>&1
timeout 600 python tools/curve_modes.py --config c1 --modes f16x2d,f16x2d::encode_w32=0 --time > $O/enc16_c1.txt 2>&1
grep -h "^\[\|Error\|error" $O/enc16_c2.txt $O/enc16_c1.txt
timeout 200 python tools/kprof.py --precision f16x2h --opt encode_w32=0 2>&1 | grep -E "^==|encode_gemm" | sed 's/info=.*//'
