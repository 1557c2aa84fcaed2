#!/bin/bash
# round-4 (second session) call 1: the driver's two exact commands at HEAD; on a fault the kernel is pinned by name
# (AMD_SERIALIZE_KERNEL=3 + ShaderName log); when green: the driver's bench line + rocprofv3 kernel stats + per-kernel tables.
O=gpurun_out/${1:-r4e}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; nproc) > $O/host.txt 2>&1
timeout 300 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e
e.smoke(); print("__SMOKE_OK__")' > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/rc.txt
if ! grep -q __SMOKE_OK__ $O/smoke.log; then
  AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=3 timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > $O/smoke_ser.out 2> $O/smoke_ser.err
  grep -a "ShaderName" $O/smoke_ser.err | tail -30 | cut -c1-300 > $O/smoke_ser_kernels.txt
  grep -a -i "fault" $O/smoke_ser.err | tail -3 >> $O/smoke_ser_kernels.txt; rm -f $O/smoke_ser.err
fi
timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -15 $O/pytest.log | cut -c1-300 > $O/pytest_tail.txt
if grep -q "pytest rc=0" $O/rc.txt; then
  timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?" >> $O/rc.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-fit --no-fp32 > $O/bench_under_rocprof.json 2> $O/trace.err
  python3 tools/rocprof_summary.py $O/trace/t_results.db > $O/kernel_stats.md 2>> $O/trace.err
  python3 tools/trace_gaps.py $O/trace/t_results.db > $O/trace_gaps.txt 2>&1
  rm -rf $O/trace
  timeout 200 python3 tools/kprof.py --precision bf16x3 > $O/kprof_x3.txt 2>&1
  timeout 200 python3 tools/kprof.py --precision bf16 > $O/kprof_bf16.txt 2>&1
else
  f=$(grep -a -o "tests/test_hip_[a-z_]*\.py" $O/pytest.log | tail -1)
  AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=3 timeout 600 python3 -m pytest ${f:-tests/test_hip_cli.py} -x -v -s -m gpu -p no:cacheprovider > $O/pytest_ser.out 2> $O/pytest_ser.err
  grep -a "ShaderName" $O/pytest_ser.err | tail -30 | cut -c1-300 > $O/pytest_ser_kernels.txt
  tail -20 $O/pytest_ser.out >> $O/pytest_ser_kernels.txt; rm -f $O/pytest_ser.err
fi
cat $O/rc.txt; cat $O/host.txt; tail -6 $O/smoke.log; cat $O/pytest_tail.txt; head -c 600 $O/bench_driver.json 2>/dev/null
