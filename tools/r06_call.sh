#!/bin/bash
# round 6, GPU call 1: native collective on one rank, f16 shadow encode fix, c3 modes against the oracle envelope
mkdir -p gpurun_out/r06c1
O=gpurun_out/r06c1
timeout 900 python -m pytest tests/test_hip_dp.py -k native tests/test_hip_f16.py -q -x > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/tests.txt
timeout 600 python tools/curve_modes.py --config c3 --modes f16x2,f16x2:343,f16x3,bf16x3,fp32 --time > $O/curve_c3.txt 2>&1
timeout 300 python tools/dp_step_breakdown.py > $O/dp_step_breakdown.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench.err
timeout 300 python bench.py --gpus 1 --force-exchange --buckets 4 --no-cpu-baseline --no-fit --no-fp32 > $O/bench_c2_exch4.json 2>> $O/bench.err
tail -5 $O/tests.txt; cat $O/curve_c3.txt; cat $O/dp_step_breakdown.txt | tail -12
