#!/bin/bash
# A/B: decode on a side stream beside the Gram -> miner chain (plan option "overlap"), interleaved, c2 and c1/c3
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
  for ov in 0 1; do
    timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fit --no-fp32 --no-roofline --option overlap=$ov > $O/ov_${ov}_$rep.json 2> $O/ov.err || tail -3 $O/ov.err
    python -c "
import json; d=json.load(open('$O/ov_${ov}_$rep.json')); print('c2 overlap=$ov rep $rep', round(d['value']), round(1e3*d['ms_per_step'],1), d['final_losses']['cost'])"
  done
done
for cfg in c1 c3 c4; do
  for ov in 0 1; do
    timeout 200 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-fit --no-fp32 --no-roofline --option overlap=$ov > $O/ov_${cfg}_$ov.json 2> $O/ov.err || tail -3 $O/ov.err
    python -c "
import json; d=json.load(open('$O/ov_${cfg}_$ov.json')); print('$cfg overlap=$ov', round(d['value']), round(1e3*d['ms_per_step'],1), d['final_losses']['cost'])"
  done
done
