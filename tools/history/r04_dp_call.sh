#!/bin/bash
# round-4: the data-parallel exchange forms of the split mode on one GPU (two ranks over gloo, one-rank RCCL breakdown)
O=gpurun_out/${1:-r4v}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python3 -m pytest tests/test_hip_dp.py tests/test_hip_cli.py -q -m gpu -p no:cacheprovider -k "dp or rank or parallel or split" > $O/pytest_dp.log 2>&1; echo "pytest_dp rc=$?" > $O/rc.txt
tail -5 $O/pytest_dp.log | cut -c1-200 >> $O/rc.txt
timeout 200 python3 tools/dp_step_breakdown.py --precision bf16x3 2>&1 | grep -a -v "amdgpu.ids\|Librccl\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname" > $O/dp_x3.txt
timeout 200 python3 bench.py --gpus 1 --steps 100 --warmup 10 --force-exchange --no-cpu-baseline --no-fit --no-fp32 --no-roofline 2> $O/bench_fx.err | grep -a "^{" > $O/bench_force_exchange.json
cat $O/rc.txt; cat $O/dp_x3.txt; head -c 400 $O/bench_force_exchange.json
