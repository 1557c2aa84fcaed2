#!/bin/bash
# dW kernel of the split-bf16 step under the compile-time probes (DAE_DW_PROBE bits: 1 one K tile, 16 no operand stream, 32 consumers idle,
# 64 fragment reads without MFMAs, 128 MFMAs without fragment reads); results of a probe run are numerically meaningless, only dw_gemm's time counts
O=gpurun_out/${1:-r4h}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
B=dae_rnn_news_recommendation_amd/csrc/build
for d in 0 1 16 32 64 128; do
  for pair in 1 0; do
    lib=""; [ $d != 0 ] && lib="--lib $B/libdae_probe_DAE_DW_PROBE=$d.so"
    timeout 100 python3 tools/kprof.py --precision bf16x3 --opt dw_pair=$pair --tag "probe$d pair$pair" $lib 2>&1 | grep -a "==\|dw_gemm" | cut -c1-130 | tr '\n' ' ' >> $O/probe.txt
    echo >> $O/probe.txt
  done
done
timeout 100 python3 tools/kprof.py --precision bf16 --tag "bf16 probe0" 2>&1 | grep -a "==\|dw_gemm" | cut -c1-130 | tr '\n' ' ' >> $O/probe.txt; echo >> $O/probe.txt
for d in 1 16 32 64 128; do
  timeout 100 python3 tools/kprof.py --precision bf16 --tag "bf16 probe$d" --lib $B/libdae_probe_DAE_DW_PROBE=$d.so 2>&1 | grep -a "==\|dw_gemm" | cut -c1-130 | tr '\n' ' ' >> $O/probe.txt; echo >> $O/probe.txt
done
cat $O/probe.txt
