#!/bin/bash
# round-3 GPU pass 4: lane-grid miner -- miner parity tests, per-kernel profile, bench line
set -u
O=gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_hip_kernels.py -q -x -k "batch_all or triplet or miner or pos_only" > $O/tests_miner.log 2>&1
tail -5 $O/tests_miner.log | cut -c1-300
timeout 500 python -m pytest tests/test_hip_step.py -q --maxfail=10 -k "batch_all or miner or full_shape or sorted or class" > $O/tests_step.log 2>&1
tail -8 $O/tests_step.log | cut -c1-300
for o in "" "--unsorted" "--precision fp32"; do
  timeout 200 python tools/kprof.py $o >> $O/kprof.txt 2>&1
done
grep -v amdgpu.ids $O/kprof.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fit', {k:v.get('samples_per_s') for k,v in d['fit'].items() if isinstance(v,dict)}, 'fp32', d['fp32']['value'])
PY
