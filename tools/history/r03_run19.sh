#!/bin/bash
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
timeout 250 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fit --no-fp32 > $O/b19_c2.json 2> $O/b19.err || tail -3 $O/b19.err
python -c "
import json; d=json.load(open('$O/b19_c2.json')); print('c2', round(d['value']), round(1e3*d['ms_per_step'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})"
timeout 250 python bench.py --config c4 --steps 50 --warmup 5 --no-cpu-baseline --no-fit --no-fp32 > $O/b19_c4.json 2> $O/b19.err || tail -3 $O/b19.err
python -c "
import json; d=json.load(open('$O/b19_c4.json')); print('c4', round(d['value']), round(1e3*d['ms_per_step'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})"
timeout 120 ./tools/lds_stream_ubench > $O/lds_stream_ubench.txt 2>&1
grep -A14 "^stores" $O/lds_stream_ubench.txt
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_step.py -q -x -k "gemm or dw or gram or step or encode or dense" > $O/tests19.log 2>&1
tail -5 $O/tests19.log | cut -c1-300
