#!/bin/bash
# round-4 evidence call (no code change behind it): SQ counters of the c2 step in both modes (MFMA busy share, effective clock), the data-parallel
# step form without communication in both modes, HBM counters of the dense-input config
O=gpurun_out/${1:-r4y}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
timeout 200 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_x3 -o s -- python3 tools/run_steps.py 20 batch_all c2 bf16x3 > /dev/null 2> $O/sq.err
python3 tools/pmc_summary.py $(find $O/sq_x3 -name "s_results.db") > $O/pmc_sq_x3.md 2>> $O/sq.err; rm -rf $O/sq_x3
timeout 200 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_bf -o s -- python3 tools/run_steps.py 20 batch_all c2 bf16 > /dev/null 2>> $O/sq.err
python3 tools/pmc_summary.py $(find $O/sq_bf -name "s_results.db") > $O/pmc_sq_bf16.md 2>> $O/sq.err; rm -rf $O/sq_bf
timeout 200 python3 tools/dp_step_breakdown.py --precision bf16x3 2>&1 | grep -a -v amdgpu.ids > $O/dp_x3.txt
timeout 200 python3 tools/dp_step_breakdown.py --precision bf16 --grad-dtype bf16 2>&1 | grep -a -v amdgpu.ids > $O/dp_bf16.txt
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f4 -o f -- python3 tools/run_steps.py 10 batch_all c4 bf16x3 > /dev/null 2>> $O/sq.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w4 -o w -- python3 tools/run_steps.py 10 batch_all c4 bf16x3 > /dev/null 2>> $O/sq.err
python3 tools/pmc_summary.py $(find $O/f4 -name "f_results.db") $(find $O/w4 -name "w_results.db") > $O/pmc_counters_c4.md 2>> $O/sq.err; rm -rf $O/f4 $O/w4
grep -a "MFMA_BUSY\|GRBM_GUI" $O/pmc_sq_x3.md | cut -c1-150; cat $O/dp_x3.txt | tail -12; tail -3 $O/dp_bf16.txt; wc -l $O/pmc_counters_c4.md
