#!/bin/bash
# A/B of the dense gather's tile shape (plan option gather_tile) on c4 + its parity test
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "
from dae_rnn_news_recommendation_amd.engine import Engine
e = Engine(8, 4, 8); print('plan ok')" || { echo "plan creation failed"; exit 1; }
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "gather_dense" > $O/tests20.log 2>&1
tail -3 $O/tests20.log | cut -c1-300
for rep in 1 2; do
for v in 0 1 2 3; do
  timeout 250 python bench.py --config c4 --steps 30 --warmup 5 --no-cpu-baseline --no-fit --no-fp32 --option gather_tile=$v > $O/b20_$v.json 2> $O/b20.err || tail -3 $O/b20.err
  python -c "
import json; d=json.load(open('$O/b20_$v.json')); k=d['kernels']; print('c4 gather_tile=$v rep $rep: step', round(1e3*d['ms_per_step'],1), 'gather', round(k['gather']['avg_us'],1), 'frac', round(k['gather']['frac'],3))"
done
done
