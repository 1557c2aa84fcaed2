#!/bin/bash
# round-3 GPU pass 1: full GPU suite, curve report, option A/Bs, bench
set -u
O=gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests.log 2>&1
tail -40 $O/tests.log
timeout 400 python tools/curve_report.py > $O/curve.txt 2>&1
for o in "" "--opt dw_sparse=0" "--opt encode_w32=0" "--opt encode_w32_cols=128" "--opt fused_opt=0" "--opt dw_sparse=0 --opt encode_w32=0" "--opt dw_sparse=0 --opt encode_w32=0 --opt fused_opt=0"; do
  timeout 200 python tools/kprof.py $o >> $O/kprof.txt 2>&1
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/curve.txt | cut -c1-400
cat $O/kprof.txt | grep -v amdgpu.ids
