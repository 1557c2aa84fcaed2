#!/bin/bash
# round 6: the whole GPU suite + smoke at the final HEAD
mkdir -p gpurun_out/r06c41
O=gpurun_out/r06c41
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 > $O/smoke.txt; cat $O/smoke.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
