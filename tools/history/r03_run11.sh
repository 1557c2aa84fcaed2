#!/bin/bash
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_step.py -q -x -k "large_dense_shape" > $O/tests_w8dw.log 2>&1
tail -12 $O/tests_w8dw.log | cut -c1-400
timeout 300 python bench.py --config c4 --steps 50 --warmup 5 --no-cpu-baseline --no-fit --no-fp32 > $O/bench_c4.json 2> $O/c4.err
python -c "
import json; d=json.load(open('gpurun_out/r3t/bench_c4.json')); print('c4', round(d['value']), round(1e3*d['ms_per_step'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})"
