#!/bin/bash
# round 6: the c3 envelope ratios of the modes at the final HEAD (the fused-stage Gram changes D's low bits in every split mode)
mkdir -p gpurun_out/r06c43
timeout 900 python tools/curve_modes.py --config c3 --modes fp32,bf16x3,f16x2h,f16x3,f16x2,f16x2:87 --time > gpurun_out/r06c43/curve_c3.txt 2>&1
grep -h "^\[\|Error" gpurun_out/r06c43/curve_c3.txt | sed 's/; ae max[^;]*;/;/'
