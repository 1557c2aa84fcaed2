#!/bin/bash
# Quick per-kernel profile of the training step on the GPU box: rocprofv3 kernel trace + SQ / traffic PMC passes of tools/run_steps.py
# usage: bash tools/quick_prof.sh <tag> [strategy]
set -u
TAG=${1:-q}; STRAT=${2:-batch_all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python tools/run_steps.py 60 $STRAT > /dev/null 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $OUT/pmc_sq -o s -- python tools/run_steps.py 20 $STRAT > /dev/null 2> $OUT/pmc.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_sq2 -o s2 -- python tools/run_steps.py 20 $STRAT > /dev/null 2>> $OUT/pmc.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python tools/run_steps.py 20 $STRAT > /dev/null 2>> $OUT/pmc.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python tools/run_steps.py 20 $STRAT > /dev/null 2>> $OUT/pmc.err
python tools/rocprof_summary.py $OUT/trace/t_results.db > $OUT/kernel_stats.md
python tools/pmc_summary.py $OUT/pmc_sq/s_results.db $OUT/pmc_sq2/s2_results.db $OUT/pmc_fetch/f_results.db $OUT/pmc_write/w_results.db > $OUT/pmc_counters.md
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write
head -16 $OUT/kernel_stats.md
