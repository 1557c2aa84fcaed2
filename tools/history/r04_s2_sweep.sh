#!/bin/bash
# split-K sweeps of the split-bf16 step (tools/kprof.py): Gram slices, dh slices
O=gpurun_out/${1:-r4i}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { timeout 100 python3 tools/kprof.py --precision bf16x3 "$@" 2>&1 | grep -a "==\|gram\|miner\|dh_gemm\|dh_finish" | cut -c1-150 | tr '\n' ' ' >> $O/sweep.txt; echo >> $O/sweep.txt; }
run --tag base
for g in 1 2 8; do run --gram-splits $g --tag "gram$g"; done
for s in 4 6 12 16; do run --enc-splits $s --tag "dh$s"; done
run --tag base2
cat $O/sweep.txt
