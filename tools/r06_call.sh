#!/bin/bash
# round 6: the whole GPU suite + smoke at HEAD (register-carry decode loops, fused-stage Gram, upload-ahead)
mkdir -p gpurun_out/r06c35
O=gpurun_out/r06c35
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/smoke.txt; cat $O/smoke.txt
