#!/bin/bash
# round 6: epoch arrays uploaded one epoch ahead on a copy stream (feeder thread): fit-level tests + the driver-form bench line, A/B against HEAD~ by env switch is not possible -> compare with call 27's numbers of the same mode
mkdir -p gpurun_out/r06c32
O=gpurun_out/r06c32
timeout 1500 python -m pytest tests/test_hip_fit.py tests/test_hip_golden_graph.py tests/test_hip_cli.py tests/test_hip_dp.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('driver form', d['value'], d['ms_per_step'], 'long', d.get('long_run',{}).get('ms_per_step'), 'fit', {k:(v.get('samples_per_s') if isinstance(v,dict) else v) for k,v in d.get('fit',{}).items()})" >> $O/bench.txt
done
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('300 steps', d['value'], d['ms_per_step'], 'fit', {k:(v.get('samples_per_s') if isinstance(v,dict) else v) for k,v in d.get('fit',{}).items()})" >> $O/bench.txt
timeout 300 python tools/region_trace.py --steps 20 --warmup 5 > $O/region_trace.txt 2>&1
cat $O/bench.txt; tail -12 $O/region_trace.txt
