#!/bin/bash
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
timeout 250 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fit --no-fp32 > $O/b23_c2.json 2> $O/b23.err || tail -3 $O/b23.err
python -c "
import json; d=json.load(open('$O/b23_c2.json')); print('c2', round(d['value']), round(1e3*d['ms_per_step'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, d['final_losses']['cost'])"
timeout 600 python -m pytest tests/test_hip_step.py tests/test_hip_kernels.py tests/test_hip_fit.py -q -x -k "not large_dense and (step or stats or tail or bias or fit)" > $O/tests23.log 2>&1
tail -4 $O/tests23.log | cut -c1-300
