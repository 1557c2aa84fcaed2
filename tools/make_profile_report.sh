#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes -> traffic JSON -> bench JSON lines of the five named configs -> rocprofv3 kernel
# trace -> tool outputs, all into gpurun_out/<tag>/;  tools/build_profile_summary.py turns that directory into the committed
# profiles/<round>_* set (run it again locally on the merged gpurun_out/<tag>/).
# usage: bash tools/make_profile_report.sh r02
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout 400"
python -c "import bench; print(bench.source_hash())" > $OUT/source_hash.txt
PMC="rocprofv3 --kernel-trace --pmc"
$T $PMC FETCH_SIZE -d $OUT/pmc_fetch -o f -- python tools/run_steps.py 20 > $OUT/run_steps.txt 2> $OUT/pmc.err
$T $PMC WRITE_SIZE -d $OUT/pmc_write -o w -- python tools/run_steps.py 20 > /dev/null 2>> $OUT/pmc.err
$T $PMC SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $OUT/pmc_sq -o s -- python tools/run_steps.py 20 > /dev/null 2>> $OUT/pmc.err
$T $PMC SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_sq2 -o s2 -- python tools/run_steps.py 20 > /dev/null 2>> $OUT/pmc.err
$T $PMC FETCH_SIZE -d $OUT/pmc_fetch4 -o f4 -- python tools/run_steps.py 10 batch_all c4 > /dev/null 2>> $OUT/pmc.err
$T $PMC WRITE_SIZE -d $OUT/pmc_write4 -o w4 -- python tools/run_steps.py 10 batch_all c4 > /dev/null 2>> $OUT/pmc.err
for cs in "1 none" "3 batch_hard"; do set -- $cs   # c1 / c3 run other precision modes than c2 (auto per strategy): their own passes
  $T $PMC FETCH_SIZE -d $OUT/pmc_fetch$1 -o f$1 -- python tools/run_steps.py 20 $2 c2 > /dev/null 2>> $OUT/pmc.err
  $T $PMC WRITE_SIZE -d $OUT/pmc_write$1 -o w$1 -- python tools/run_steps.py 20 $2 c2 > /dev/null 2>> $OUT/pmc.err
  python tools/pmc_summary.py $OUT/pmc_fetch$1/f$1_results.db $OUT/pmc_write$1/w$1_results.db > $OUT/pmc_counters_c$1.md
  rm -rf $OUT/pmc_fetch$1 $OUT/pmc_write$1
done
$T $PMC FETCH_SIZE -d $OUT/pmc_fetch5 -o f5 -- python tools/run_steps.py 20 explicit c5 > /dev/null 2>> $OUT/pmc.err
$T $PMC WRITE_SIZE -d $OUT/pmc_write5 -o w5 -- python tools/run_steps.py 20 explicit c5 > /dev/null 2>> $OUT/pmc.err
python tools/pmc_summary.py $OUT/pmc_fetch/f_results.db $OUT/pmc_write/w_results.db $OUT/pmc_sq/s_results.db $OUT/pmc_sq2/s2_results.db > $OUT/pmc_counters.md
python tools/pmc_summary.py $OUT/pmc_fetch4/f4_results.db $OUT/pmc_write4/w4_results.db > $OUT/pmc_counters_c4.md
python tools/pmc_summary.py $OUT/pmc_fetch5/f5_results.db $OUT/pmc_write5/w5_results.db > $OUT/pmc_counters_c5.md
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch4 $OUT/pmc_write4 $OUT/pmc_fetch5 $OUT/pmc_write5
# the traffic file bench.py quotes (same kernels, same box, minutes apart)
python tools/build_profile_summary.py $OUT $TAG --traffic-only > /dev/null
$T python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err      # exactly the driver's command
$T python bench.py --config c2 --steps 300 --warmup 30 > $OUT/bench_c2.json 2>> $OUT/bench.err
$T python bench.py --config c1 --steps 300 --warmup 30 > $OUT/bench_c1.json 2>> $OUT/bench.err
$T python bench.py --config c3 --steps 300 --warmup 30 > $OUT/bench_c3.json 2>> $OUT/bench.err
$T python bench.py --config c5 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c5.json 2>> $OUT/bench.err
$T python bench.py --config c4 --steps 100 --warmup 10 > $OUT/bench_c4.json 2>> $OUT/bench.err
$T rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-fit --no-fp32 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
python tools/rocprof_summary.py $OUT/trace/t_results.db > $OUT/kernel_stats.md
python tools/trace_gaps.py $OUT/trace/t_results.db > $OUT/trace_gaps.txt 2>&1
rm -rf $OUT/trace
bash tools/pmc_c4_sq.sh $TAG > /dev/null 2>&1   # SQ counters of the c4 GEMM kernels (own PMC pass) -> $OUT/pmc_counters_c4_sq.md
$T python tools/kprof.py --stamps --precision f16x2h > $OUT/kprof.txt 2>/dev/null      # the product default for batch_all, per kernel
$T python tools/kprof.py --stamps --precision f16x2d --strategy none >> $OUT/kprof.txt 2>/dev/null      # ... for strategy none
$T python tools/kprof.py --stamps --precision f16x2h --strategy batch_hard >> $OUT/kprof.txt 2>/dev/null      # ... for batch_hard
$T python tools/kprof.py --stamps --precision f16x2 >> $OUT/kprof.txt 2>/dev/null       # round 5's default (holds 20 steps, not 100)
$T python tools/kprof.py --stamps --precision bf16x3 >> $OUT/kprof.txt 2>/dev/null      # the split-bf16 mode
$T python tools/kprof.py --stamps --precision bf16 >> $OUT/kprof.txt 2>/dev/null
[ -f dae_rnn_news_recommendation_amd/libdae_mp4.so ] && $T python tools/miner_timeline.py --lib dae_rnn_news_recommendation_amd/libdae_mp4.so > $OUT/miner_timeline.txt 2> $OUT/miner_timeline.err
$T python tools/dp_step_breakdown.py > $OUT/dp_step_breakdown.txt 2>/dev/null
nproc > $OUT/host.txt; lscpu | grep "Model name" >> $OUT/host.txt; rocminfo | grep -E "gfx|Compute Unit" | head -4 >> $OUT/host.txt
ls -la $OUT
