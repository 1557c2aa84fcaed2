#!/bin/bash
# round 6: the three event forms of dae_plan_profile (host wait / queued markers / dispatch stamps) against rocprofv3
mkdir -p gpurun_out/r06c46; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_profile.py -x -q -m gpu 2>&1 | tail -8
for q in "" "--queued" "--stamps"; do
  timeout 200 python tools/kprof.py --precision f16x2h $q 2>/dev/null >> gpurun_out/r06c46/kprof_ab.txt
done
cat gpurun_out/r06c46/kprof_ab.txt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r06c46/trace -o t -- python tools/kprof.py --precision f16x2h --stamps > gpurun_out/r06c46/kprof_under_rocprof.txt 2>/dev/null
python tools/rocprof_summary.py gpurun_out/r06c46/trace/t_results.db | head -16; rm -rf gpurun_out/r06c46/trace
cat gpurun_out/r06c46/kprof_under_rocprof.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06c46/bench.json 2> gpurun_out/r06c46/bench.err; echo "rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r06c46/bench.json').readline())
print(d['value'], d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'], d['profiled_step_us'])
print({k:round(v.get('avg_us'),1) for k,v in d['kernels'].items()})
"
