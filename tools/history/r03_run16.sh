#!/bin/bash
# dW K-loop probes (compile-time twins of the library) + the L2 -> LDS stream micro-benchmark
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
B=dae_rnn_news_recommendation_amd/csrc/build
for rep in 1 2; do
for v in base 16 32 48 64 128 1; do
  if [ $v = base ]; then lib=dae_rnn_news_recommendation_amd/libdae_hip.so; else lib=$B/libdae_probe_DAE_DW_PROBE=$v.so; fi
  timeout 200 python tools/bench_with_lib.py $lib --steps 40 --warmup 10 --no-cpu-baseline --no-fit --no-fp32 > $O/dwp_$v.json 2> $O/dwp.err || tail -3 $O/dwp.err
  python -c "
import json; d=json.load(open('$O/dwp_$v.json')); k=d['kernels']; print('probe $v rep $rep: step', round(1e3*d['ms_per_step'],1), 'dw_gemm', round(k['dw_gemm']['avg_us'],1))"
done
done
timeout 120 ./tools/lds_stream_ubench > $O/lds_stream_ubench.txt 2>&1
tail -32 $O/lds_stream_ubench.txt
