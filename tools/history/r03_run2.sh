#!/bin/bash
# round-3 GPU pass 2: step tests (new dW bit-image path), A/B kprof, bench
set -u
O=gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_step.py tests/test_hip_fit.py tests/test_hip_golden_graph.py tests/test_hip_configs.py -q --maxfail=25 > $O/tests.log 2>&1
tail -30 $O/tests.log | cut -c1-300
for o in "" "--opt dw_bits=0" "--opt encode_w32=0" "--opt dw_bits=0 --opt encode_w32=0" "--strategy none" "--strategy batch_hard"; do
  timeout 200 python tools/kprof.py $o >> $O/kprof.txt 2>&1
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
cat $O/kprof.txt | grep -v amdgpu.ids
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fit', {k:v.get('samples_per_s') for k,v in d['fit'].items() if isinstance(v,dict)}, 'fp32', d['fp32'])
PY
