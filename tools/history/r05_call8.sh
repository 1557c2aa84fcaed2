#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
./tools/tr_probe > gpurun_out/r05_tr_probe.txt 2>&1; tail -1 gpurun_out/r05_tr_probe.txt
timeout 1500 python -m pytest tests/test_hip_f16.py tests/test_hip_step.py tests/test_hip_configs.py tests/test_hip_full_curve.py -q -x -m gpu -k "not class_range_path and not snake_packing and not dispatch_order" > gpurun_out/r05_call8_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_call8_tests.txt
tail -4 gpurun_out/r05_call8_tests.txt
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 $2 > gpurun_out/r05_call8_$1.json 2>> gpurun_out/r05_call8.log
  python - "gpurun_out/r05_call8_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("kernels",{})
print(sys.argv[1].split("call8_")[1], d["dtype"], "us/step %.1f long %.1f" % (1e3*(d.get("short_run") or d)["ms_per_step"], 1e3*(d.get("long_run") or {}).get("ms_per_step",0)), " ".join("%s=%.1f" % (n[:6], v["avg_us"]) for n,v in k.items()))
PY
}
run skip1 "--config c2"
run skip0 "--config c2 --option pad_skip=0"
run skip1b "--config c2"
run skip0b "--config c2 --option pad_skip=0"
run c4 "--config c4"
