#!/bin/bash
# round 6: f16 test file at the committed register-carry loops (x3 / c2), then the rest of the GPU suite
mkdir -p gpurun_out/r06c31
O=gpurun_out/r06c31
timeout 900 python -m pytest tests/test_hip_f16.py -q -x 2>&1 | tail -12 > $O/tests_f16.txt; cat $O/tests_f16.txt
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_hip_f16.py 2>&1 | tail -8 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
