#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_pmcq; mkdir -p $OUT; export TMPDIR=/tmp
PMC="rocprofv3 --kernel-trace --pmc"
timeout 400 $PMC FETCH_SIZE -d $OUT/pmc_fetch -o f -- python tools/run_steps.py 20 > $OUT/run_steps.txt 2> $OUT/pmc.err
timeout 400 $PMC WRITE_SIZE -d $OUT/pmc_write -o w -- python tools/run_steps.py 20 > /dev/null 2>> $OUT/pmc.err
python tools/pmc_summary.py $OUT/pmc_fetch/f_results.db $OUT/pmc_write/w_results.db > $OUT/pmc_counters.md
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_counters.md | grep -v "at::\|Cijk\|elementwise"
