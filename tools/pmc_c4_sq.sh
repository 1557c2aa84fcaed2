#!/bin/bash
# PMC pass for the dense-input config c4: MFMA busy share of the 256 x 256-tile GEMM kernels (own run: counters + kernel trace only)
set -u
O=gpurun_out/${1:-r03}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $O/pmc_sq4 -o s4 -- python tools/run_steps.py 10 batch_all c4 > /dev/null 2> $O/pmc4.err
python tools/pmc_summary.py $O/pmc_sq4/s4_results.db > $O/pmc_counters_c4_sq.md
rm -rf $O/pmc_sq4
grep -E "gemm_nt_w8|gemm_dw_opt|gemm_decode" $O/pmc_counters_c4_sq.md | grep -E "MFMA_BUSY|GRBM_GUI|SQ_BUSY_CU" | cut -c1-200
