#!/bin/bash
# same-box A/B of the gemm_nt_pc slab epilogue (dae_set_glds(-9) = former dword stores), interleaved, both precision modes
O=gpurun_out/${1:-r4k}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { timeout 100 python3 tools/kprof.py "$@" 2>&1 | grep -a "==\|gram\|dh_gemm\|miner" | cut -c1-140 | tr '\n' ' ' >> $O/ab.txt; echo >> $O/ab.txt; }
for rep in 1 2 3; do
  run --precision bf16x3 --tag "x3 vec"
  run --precision bf16x3 --glds -9 --tag "x3 dword"
done
run --precision bf16 --tag "bf16 vec"; run --precision bf16 --glds -9 --tag "bf16 dword"
run --precision bf16 --tag "bf16 vec"; run --precision bf16 --glds -9 --tag "bf16 dword"
cat $O/ab.txt
