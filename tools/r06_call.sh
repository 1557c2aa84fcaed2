#!/bin/bash
# round 6: c1 over the CLI's default 50 epochs = 500 steps -- which modes hold 1e-4 to the end?
mkdir -p gpurun_out/r06c20
O=gpurun_out/r06c20
timeout 900 python tools/curve_modes.py --config c1 --epochs 50 --modes f16x2d,f16x2,f16x2:173,bf16x3,f16x3 > $O/curve_c1_e50.txt 2>&1
grep -h "^\[\|Error" $O/curve_c1_e50.txt
