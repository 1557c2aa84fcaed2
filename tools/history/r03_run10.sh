#!/bin/bash
set -u
O=gpurun_out/r3r
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python bench.py --force-exchange --steps 100 --warmup 10 --no-cpu-baseline --no-fit --no-fp32 > $O/bench_force_exchange.json 2> $O/fe.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3r/bench_force_exchange.json'))
print('force-exchange (1-rank RCCL): value', round(d['value']), 'us/step', round(1e3*d['ms_per_step'],1), 'collective_us', d.get('collective_us'), 'local_step_us', d.get('local_step_us'), 'exposed_us', d.get('exposed_us'))
print(d['config']['collective'])
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --backend gloo --single-device --steps 30 --warmup 5 --no-cpu-baseline --no-fit --no-fp32 > $O/bench_2rank_gloo.json 2> $O/g2.err
tail -2 $O/g2.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3r/bench_2rank_gloo.json').read().strip().splitlines()[-1])
print('2 ranks on one GPU (gloo): value', round(d['value']), 'us/step', round(1e3*d['ms_per_step'],1), 'n_gpus', d['n_gpus'], 'final cost', d['final_losses']['cost'])
PY
