#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 --no-roofline $2 > gpurun_out/r05_call13_$1.json 2>> gpurun_out/r05_call13.log
  python - "gpurun_out/r05_call13_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split("call13_")[1], d["dtype"], "us/step %.1f long %.1f" % (1e3*(d.get("short_run") or d)["ms_per_step"], 1e3*(d.get("long_run") or {}).get("ms_per_step",0)), "cost", d["final_losses"]["cost"])
PY
}
run base "--config c2"
run ov1 "--config c2 --option overlap=1"
run ov2 "--config c2 --option overlap=2"
run ov3 "--config c2 --option overlap=3"
run base2 "--config c2"
run ov3b "--config c2 --option overlap=3"
run ov2b "--config c2 --option overlap=2"
