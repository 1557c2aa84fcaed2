#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_f16.py -q -x -m gpu > gpurun_out/r05_call12_tests.txt 2>&1; tail -3 gpurun_out/r05_call12_tests.txt
timeout 600 python tools/curve_terms.py --precision f16x2 --terms 5 2>&1 | grep x3_terms
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 $2 > gpurun_out/r05_call12_$1.json 2>> gpurun_out/r05_call12.log
  python - "gpurun_out/r05_call12_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("kernels",{})
print(sys.argv[1].split("call12_")[1], d["dtype"], "us/step %.1f long %.1f" % (1e3*(d.get("short_run") or d)["ms_per_step"], 1e3*(d.get("long_run") or {}).get("ms_per_step",0)), " ".join("%s=%.1f" % (n[:6], v["avg_us"]) for n,v in k.items()))
PY
}
run pair0 "--config c2 --option decode_pair=0"
run pair1 "--config c2 --option decode_pair=1"
run pair0b "--config c2 --option decode_pair=0"
run pair1b "--config c2 --option decode_pair=1"
run c1_pair1 "--config c1 --option decode_pair=1"
run c1_pair0 "--config c1 --option decode_pair=0"
