#!/bin/bash
# round 6: fp16-storage lo-term masks on c3 against the envelope gate (is there a cheaper passing mode than bf16x3 for batch_hard?)
mkdir -p gpurun_out/r06c22
timeout 900 python tools/curve_modes.py --config c3 --modes f16x2h,f16x2:39,f16x2:111,f16x2:47,bf16x3 --time > gpurun_out/r06c22/curve_c3_masks2.txt 2>&1
grep -h "^\[\|Error" gpurun_out/r06c22/curve_c3_masks2.txt | sed 's/; ae max[^;]*;/;/'
