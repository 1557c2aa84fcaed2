#!/bin/bash
# round 6: the long-curve tests incl. the 500-step c1 / c5 cases
mkdir -p gpurun_out/r06c21
timeout 1500 python -m pytest tests/test_hip_long_curves.py -q -s 2>&1 | grep -E "long curve\]|passed|failed|FAILED|Error|^E " > gpurun_out/r06c21/long_curves.txt; cat gpurun_out/r06c21/long_curves.txt
