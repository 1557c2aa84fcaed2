#!/bin/bash
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
timeout 250 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fit --no-fp32 > $O/b18_c2.json 2> $O/b18.err || tail -3 $O/b18.err
python -c "
import json; d=json.load(open('$O/b18_c2.json')); print('c2', round(d['value']), round(1e3*d['ms_per_step'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, d['final_losses'])"
timeout 900 python -m pytest tests/test_hip_step.py tests/test_hip_dp.py -q -x > $O/tests18.log 2>&1
tail -5 $O/tests18.log | cut -c1-300
