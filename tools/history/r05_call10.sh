#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/c4_terms.py 261,5,257 > gpurun_out/r05_c4_terms.txt 2>&1; grep "c4 f16x2" gpurun_out/r05_c4_terms.txt
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 $2 > gpurun_out/r05_call10_$1.json 2>> gpurun_out/r05_call10.log
  python - "gpurun_out/r05_call10_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("kernels",{})
print(sys.argv[1].split("call10_")[1], d["dtype"], "us/step %.1f" % (1e3*(d.get("short_run") or d)["ms_per_step"]), " ".join("%s=%.1f" % (n[:6], v["avg_us"]) for n,v in k.items()))
PY
}
run c4_261 "--config c4"
run c4_5 "--config c4 --option x3_terms=5"
