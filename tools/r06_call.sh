#!/bin/bash
# round 6: run-to-run determinism of the product defaults at the full c2 shape
mkdir -p gpurun_out/r06c37
timeout 900 python -m pytest tests/test_hip_f16.py -q -x -k "run_to_run" 2>&1 | tail -6 > gpurun_out/r06c37/tests.txt; cat gpurun_out/r06c37/tests.txt
