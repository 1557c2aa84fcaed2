#!/bin/bash
# inter-kernel gaps of the c2 step from a rocprofv3 kernel trace + A/B of the optimizer fused into the dW GEMM's epilogue
set -u
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-fit --no-fp32 > /dev/null 2> $O/trace2.err
python tools/trace_gaps.py $O/trace/t_results.db > $O/trace_gaps.txt 2>&1
rm -rf $O/trace
cat $O/trace_gaps.txt
for rep in 1 2 3; do
for v in 1 0; do
  timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fit --no-fp32 --no-roofline --option fused_opt=$v > $O/fo_$v.json 2> $O/fo.err || tail -3 $O/fo.err
  python -c "
import json; d=json.load(open('$O/fo_$v.json')); print('fused_opt=$v rep $rep:', round(d['value']), 'samples/s', round(1e3*d['ms_per_step'],1), 'us/step')"
done
done 2>&1 | tee $O/fused_opt_ab.txt
