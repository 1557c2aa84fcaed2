#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
# base = 261 (W@dec 1, W@dh 4, W@enc 256); +2 h@dec, +16 h@dh, +64 h@dw, +8 d2@dh, +128 d2@dw, +32 d1@dw
timeout 1200 python tools/curve_terms.py --precision f16x2 --time --terms 261,277,325,341,263,343,269,389,293,2047,260,257 > gpurun_out/r05_curve_terms.txt 2>&1
cat gpurun_out/r05_curve_terms.txt | grep -v Warning
timeout 300 python tools/curve_terms.py --precision f16x2 --terms 261 --scale-log2 8,11,13,14 >> gpurun_out/r05_curve_terms.txt 2>&1
tail -4 gpurun_out/r05_curve_terms.txt
