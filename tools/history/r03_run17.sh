#!/bin/bash
# dW epilogue probes: which of the master-weight read / store / shadow stores costs what
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
B=dae_rnn_news_recommendation_amd/csrc/build
for rep in 1 2; do
for v in base 2 8 10 4 14 1 15; do
  if [ $v = base ]; then lib=dae_rnn_news_recommendation_amd/libdae_hip.so; else lib=$B/libdae_probe_DAE_DW_PROBE=$v.so; fi
  timeout 200 python tools/bench_with_lib.py $lib --steps 40 --warmup 10 --no-cpu-baseline --no-fit --no-fp32 > $O/dwq_$v.json 2> $O/dwq.err || tail -3 $O/dwq.err
  python -c "
import json; d=json.load(open('$O/dwq_$v.json')); k=d['kernels']; print('probe $v rep $rep: step', round(1e3*d['ms_per_step'],1), 'dw_gemm', round(k['dw_gemm']['avg_us'],1), 'decode', round(k['decode_loss']['avg_us'],1))"
done
done
