#!/bin/bash
# Run on the GPU box (via gpurun): bench JSON + rocprofv3 kernel trace + PMC passes into gpurun_out/<tag>/
# usage: bash tools/make_profile_report.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 300 --warmup 30 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --rng numpy --no-cpu-baseline --no-roofline > $OUT/bench_numpy_rng.json 2>> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --strategy none --no-cpu-baseline --no-roofline > $OUT/bench_none.json 2>> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --strategy batch_hard --no-cpu-baseline --no-roofline > $OUT/bench_batch_hard.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python tools/run_steps.py 20 > /dev/null 2> $OUT/pmc.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python tools/run_steps.py 20 > /dev/null 2>> $OUT/pmc.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $OUT/pmc_sq -o s -- python tools/run_steps.py 20 > /dev/null 2>> $OUT/pmc.err
python tools/rocprof_summary.py $OUT/trace/t_results.db > $OUT/kernel_stats.md
python tools/pmc_summary.py $OUT/pmc_fetch/f_results.db $OUT/pmc_write/w_results.db $OUT/pmc_sq/s_results.db > $OUT/pmc_counters.md
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
python tools/gemm_trace.py --nst 2 > $OUT/gemm_trace.txt 2>/dev/null
python tools/kprof.py > $OUT/kprof.txt 2>/dev/null
nproc > $OUT/host.txt; lscpu | grep "Model name" >> $OUT/host.txt; rocminfo | grep -E "gfx|Compute Unit" | head -4 >> $OUT/host.txt
ls -la $OUT
