#!/bin/bash
# round-4 (second session) call 2: the split-bf16 changes of this session -- dropped lo terms, paired dW stages -- on the GPU:
#   (1) the split-mode tests (steps, configs, the 20-step curve with and without the dropped terms, data parallel, CLI, fit)
#   (2) smoke() under the guard-page allocator (both placements)
#   (3) A/B per-kernel timings: dw_pair 0/1, all terms, overlap, plain bf16
#   (4) one bench line
O=gpurun_out/${1:-r4f}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T0=$(date +%s)
timeout 420 python3 -m pytest tests/test_hip_step.py tests/test_hip_configs.py tests/test_hip_full_curve.py tests/test_hip_dp.py -q -m gpu -p no:cacheprovider -s \
    -k "x3 or curve or config or smoke" > $O/pytest_x3.log 2>&1; echo "pytest_x3 rc=$? t=$(( $(date +%s) - T0 ))" > $O/rc.txt
grep -a "\[curve\]\|passed\|failed\|FAILED\|rel err\|Error\|error" $O/pytest_x3.log | cut -c1-260 | tail -60 > $O/pytest_x3_tail.txt
for m in end start; do
  DAE_GUARD_ALLOC=$m timeout 200 python3 -c "
import sys; sys.path.insert(0, 'tests'); import conftest
import __graft_entry__ as e; e.smoke(); print('__SMOKE_OK__')" > $O/guard_$m.log 2>&1; echo "guard_$m rc=$? t=$(( $(date +%s) - T0 ))" >> $O/rc.txt
done
DAE_GUARD_ALLOC=end timeout 240 python3 -m pytest tests/test_hip_cli.py tests/test_hip_fit.py -x -q -m gpu -p no:cacheprovider > $O/guard_cli_fit.log 2>&1; echo "guard_cli_fit rc=$? t=$(( $(date +%s) - T0 )) $(tail -1 $O/guard_cli_fit.log | cut -c1-80)" >> $O/rc.txt
K="timeout 120 python3 tools/kprof.py --precision bf16x3"
$K --tag default > $O/kprof_default.txt 2>&1
$K --opt dw_pair=0 --tag unpaired > $O/kprof_unpaired.txt 2>&1
$K --opt x3_dec_wlo=1 --opt x3_dh_hlo=1 --tag allterms > $O/kprof_allterms.txt 2>&1
$K --opt overlap=1 --tag overlap > $O/kprof_overlap.txt 2>&1
timeout 120 python3 tools/kprof.py --precision bf16 --tag bf16 > $O/kprof_bf16.txt 2>&1
echo "kprof done t=$(( $(date +%s) - T0 ))" >> $O/rc.txt
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$? t=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt; cat $O/pytest_x3_tail.txt | tail -30; tail -2 $O/guard_end.log $O/guard_start.log
grep -a "==" -A9 $O/kprof_default.txt $O/kprof_unpaired.txt $O/kprof_allterms.txt | grep -v amdgpu | cut -c1-150
grep -a "==" $O/kprof_overlap.txt $O/kprof_bf16.txt | cut -c1-150
head -c 400 $O/bench_c2.json
