#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_curves.py tests/test_hip_f16.py "tests/test_hip_step.py::test_apply_in_row_bands_equals_one_apply" "tests/test_hip_step.py::test_step_split_mode_dense_train_set_with_corrupted_csr_copy" tests/test_hip_kernels.py -q -s -m gpu -k "not test_miners_gradients" > gpurun_out/r05_call5_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_call5_tests.txt
grep "\[curve\]\|passed\|failed\|rror\|bf16x3 cost\|f16x3 cost" gpurun_out/r05_call5_tests.txt | tail -20
run() { # name, args
  timeout 600 python bench.py --no-cpu-baseline --no-fit --no-fp32 $2 > gpurun_out/r05_call5_$1.json 2>> gpurun_out/r05_call5.log
  python - "gpurun_out/r05_call5_$1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("kernels",{})
print(sys.argv[1].split("call5_")[1], d["dtype"], "us/step %.1f" % (1e3*d["ms_per_step"]), " ".join("%s=%.1f" % (n[:6], v["avg_us"]) for n,v in k.items()))
PY
}
run c2_bn64 "--config c2 --option decode_bn=64"
run c2_bn128 "--config c2 --option decode_bn=128"
run c4_auto "--config c4"
run c4_bn64 "--config c4 --option decode_bn=64"
run c4_rounds1 "--config c4 --option dw_rounds=1"
run c2_again "--config c2"
