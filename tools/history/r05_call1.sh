#!/bin/bash
# round 5, first GPU call: the fp16 build (precision f16x2) -- MFMA subnormal check, step parity, the 20-step full-shape curve, a first bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_f16.py -x -q -s -m gpu > gpurun_out/r05_call1_f16_tests.txt 2>&1
echo "f16 tests rc=$?" | tee -a gpurun_out/r05_call1_f16_tests.txt
tail -5 gpurun_out/r05_call1_f16_tests.txt
timeout 900 python -m pytest tests/test_hip_full_curve.py -q -s -m gpu -k "f16x2 or bf16x3" > gpurun_out/r05_call1_curve.txt 2>&1
echo "curve rc=$?" | tee -a gpurun_out/r05_call1_curve.txt
grep "\[curve\]\|passed\|failed" gpurun_out/r05_call1_curve.txt | tail -40
timeout 600 python bench.py --precision f16x2 --no-cpu-baseline --no-fit > gpurun_out/r05_call1_bench_f16x2.json 2> gpurun_out/r05_call1_bench_f16x2.log
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_call1_bench_f16x2.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "long", d.get("long_run"))
for k,v in d.get("kernels",{}).items(): print("  %-16s %7.1f us x %.1f" % (k, v["avg_us"], v["launches_per_step"]))
for m in ("bf16x3","fp32","bf16"): print(m, d.get(m))
PY
