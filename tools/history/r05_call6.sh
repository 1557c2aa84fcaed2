#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 $2 > gpurun_out/r05_call6_$1.json 2>> gpurun_out/r05_call6.log
  python - "gpurun_out/r05_call6_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("kernels",{})
print(sys.argv[1].split("call6_")[1], d["dtype"], "us/step %.1f long %.1f" % (1e3*d["ms_per_step"], 1e3*(d.get("long_run") or {}).get("ms_per_step",0)), " ".join("%s=%.1f" % (n[:6], v["avg_us"]) for n,v in k.items()))
PY
}
run base "--config c2"
run overlap "--config c2 --option overlap=1"
run pack0 "--config c2 --option miner_pack=0"
run pack1 "--config c2 --option miner_pack=1"
run order0 "--config c2 --option miner_order=0"
run symsep "--config c2 --option sym_in_decode=0"
run base2 "--config c2"
