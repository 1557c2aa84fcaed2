#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_curves.py -q -s -m gpu > gpurun_out/r05_call4_curves.txt 2>&1; echo "curves rc=$?" >> gpurun_out/r05_call4_curves.txt
grep "\[curve\]\|passed\|failed\|rror" gpurun_out/r05_call4_curves.txt | tail -20
for opt in "" "--option dw_rounds=16"; do
  timeout 600 python bench.py --config c4 --no-cpu-baseline --no-fit --no-fp32 $opt > gpurun_out/r05_call4_c4_${opt: -2}.json 2> gpurun_out/r05_call4_c4.log
  python - "gpurun_out/r05_call4_c4_${opt: -2}.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["dtype"], "ms/step", d["ms_per_step"], "value", d["value"])
for k,v in d.get("kernels",{}).items(): print("  %-16s %7.1f us x %.1f  %s" % (k, v["avg_us"], v["launches_per_step"], v.get("frac")))
PY
done
for cfg in c1 c3 c5; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-fit --no-fp32 > gpurun_out/r05_call4_$cfg.json 2>> gpurun_out/r05_call4_c4.log
  python - gpurun_out/r05_call4_$cfg.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["dtype"], "ms/step", d["ms_per_step"], "value", d["value"])
PY
done
