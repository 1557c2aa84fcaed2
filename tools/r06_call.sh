#!/bin/bash
# round 6: the 100-step curve of c4 (dense tf-idf, F = 50000, batch_all) in auto = f16x2h, and in f16x2 / bf16x3 for context
mkdir -p gpurun_out/r06c36
O=gpurun_out/r06c36
timeout 1200 python -m pytest tests/test_hip_long_curves.py -q -x -s -k "c4" 2>&1 | grep -E "curve\]|passed|failed|FAILED|Error|^E " | cut -c1-260 > $O/curves.txt; cat $O/curves.txt
timeout 900 python tools/curve_modes.py --config c4 --modes f16x2h,f16x2,bf16x3 --time > $O/curve_c4_modes.txt 2>&1; grep -h "^\[\|Error" $O/curve_c4_modes.txt | sed 's/; ae max[^;]*;/;/'
