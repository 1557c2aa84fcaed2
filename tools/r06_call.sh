#!/bin/bash
# round 6: cosine z buffer in the kernel's register order (16-byte coalesced accesses): equivalence + cosine tests, c5 A/B
mkdir -p gpurun_out/r06c40
O=gpurun_out/r06c40
timeout 1200 python -m pytest tests/test_hip_step.py tests/test_hip_kernels.py tests/test_hip_f16.py tests/test_hip_fit.py tests/test_hip_golden_graph.py tests/test_hip_configs.py -q -x -k "cos or explicit or triplet or c5" 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
timeout 300 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline --no-fit --no-fp32 --option cos_zstore=0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5 recompute', round(d['value']), round(d['ms_per_step']*1e3,1), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('decode_loss','cos_reduce')})" >> $O/bench.txt
timeout 300 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline --no-fit --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5 zstore   ', round(d['value']), round(d['ms_per_step']*1e3,1), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('decode_loss','cos_reduce')})" >> $O/bench.txt
done
cat $O/bench.txt
