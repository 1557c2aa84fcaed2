#!/bin/bash
# round 6: the multi-rank paths of bench.py / fit() on one GPU after the pre-warm change + the new c-config tests
mkdir -p gpurun_out/r06c19
O=gpurun_out/r06c19
timeout 1200 python -m pytest tests/test_hip_cli.py tests/test_hip_dp.py -q > $O/dp_tests.txt 2>&1; echo "rc $?" >> $O/dp_tests.txt; tail -4 $O/dp_tests.txt
timeout 300 python bench.py --gpus 1 --force-exchange --no-cpu-baseline --no-fit --no-fp32 --steps 20 --warmup 5 > $O/bench_force_exchange.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_force_exchange.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','collective_us','exposed_us','local_step_us','ranks_seen','exchange')})"
