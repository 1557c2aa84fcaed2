#!/bin/bash
# round-4 final call: the driver's two commands on the final tree, then the round's committed measurement set
#   smoke -> pytest -m gpu -> PMC passes (FETCH_SIZE / WRITE_SIZE, own runs) -> traffic JSON -> bench c2 (the driver's command) -> rocprofv3 kernel
#   stats of the same bench -> the other named configs -> per-kernel tables
O=gpurun_out/${1:-r4z}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T0=$(date +%s); stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))" >> $O/rc.txt; }
(nproc; lscpu | grep "Model name"; rocminfo | grep -E "gfx|Compute Unit" | head -4) > $O/host.txt 2>&1
python3 -c "import bench; print(bench.source_hash())" > $O/source_hash.txt
timeout 300 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e
e.smoke(); print("__SMOKE_OK__")' > $O/smoke.log 2>&1; stamp smoke $?
timeout 900 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; stamp pytest $?
tail -12 $O/pytest.log | cut -c1-300 > $O/pytest_tail.txt
PMC="timeout 200 rocprofv3 --kernel-trace --pmc"
$PMC FETCH_SIZE -d $O/pmc_fetch -o f -- python3 tools/run_steps.py 20 > $O/run_steps.txt 2> $O/pmc.err; stamp pmc_fetch $?
$PMC WRITE_SIZE -d $O/pmc_write -o w -- python3 tools/run_steps.py 20 > /dev/null 2>> $O/pmc.err; stamp pmc_write $?
python3 tools/pmc_summary.py $(find $O/pmc_fetch -name "f_results.db") $(find $O/pmc_write -name "w_results.db") > $O/pmc_counters.md 2>> $O/pmc.err
rm -rf $O/pmc_fetch $O/pmc_write
python3 tools/build_profile_summary.py $O r04 --traffic-only > /dev/null 2>> $O/pmc.err; cp profiles/r04_pmc_traffic.json $O/ 2>/dev/null
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench.err; stamp bench_c2 $?
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-fit --no-fp32 > $O/bench_under_rocprof.json 2> $O/trace.err; stamp rocprof $?
DB=$(find $O/trace -name "t_results.db" | head -1)
python3 tools/rocprof_summary.py $DB > $O/kernel_stats.md 2>> $O/trace.err
python3 tools/trace_gaps.py $DB > $O/trace_gaps.txt 2>&1
rm -rf $O/trace
timeout 200 python3 bench.py --config c1 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_c1.json 2>> $O/bench.err; stamp bench_c1 $?
timeout 200 python3 bench.py --config c3 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench.err; stamp bench_c3 $?
timeout 200 python3 bench.py --config c5 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err; stamp bench_c5 $?
(timeout 120 python3 tools/kprof.py --precision bf16x3; timeout 120 python3 tools/kprof.py --precision bf16) 2>&1 | grep -a -v amdgpu.ids > $O/kprof.txt; stamp kprof $?
timeout 300 python3 bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err; stamp bench_c4 $?
cat $O/rc.txt; tail -4 $O/smoke.log; cat $O/pytest_tail.txt | tail -4; head -c 300 $O/bench_c2.json; echo; ls $O
