#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_hip_f16.py tests/test_hip_step.py tests/test_hip_configs.py tests/test_hip_full_curve.py tests/test_hip_curves.py -q -x -m gpu -k "not class_range_path and not snake_packing and not dispatch_order" > gpurun_out/r05_call9_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_call9_tests.txt
tail -6 gpurun_out/r05_call9_tests.txt
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 $2 > gpurun_out/r05_call9_$1.json 2>> gpurun_out/r05_call9.log
  python - "gpurun_out/r05_call9_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("kernels",{})
print(sys.argv[1].split("call9_")[1], d["dtype"], "us/step %.1f long %.1f" % (1e3*(d.get("short_run") or d)["ms_per_step"], 1e3*(d.get("long_run") or {}).get("ms_per_step",0)), " ".join("%s=%.1f" % (n[:6], v["avg_us"]) for n,v in k.items()))
PY
}
run tr1 "--config c2"
run tr0 "--config c2 --option dw_tr=0"
run tr1b "--config c2"
run tr0b "--config c2 --option dw_tr=0"
run c4_tr1 "--config c4"
run c4_tr0 "--config c4 --option dw_tr=0"
run c1_tr1 "--config c1"
run c1_tr0 "--config c1 --option dw_tr=0"
