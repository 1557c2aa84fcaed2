#!/bin/bash
# round-4 call 3: (A) the whole GPU suite once (new split-mode tests included), (B) guard pages in front of every buffer + poisoned fresh
# memory, (C) forced CU counts
O=gpurun_out/r4d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider -rf > $O/A_pytest.log 2>&1; echo "A pytest rc=$?" > $O/summary.txt
tail -40 $O/A_pytest.log | cut -c1-300 >> $O/summary.txt
export DAE_GUARD_POISON=1
bash tools/r04_guard.sh $O/B "start" "" > $O/B.log 2>&1
unset DAE_GUARD_POISON
bash tools/r04_guard.sh $O/C "" "64 32" tests/test_hip_step.py tests/test_hip_cli.py tests/test_hip_fit.py tests/test_hip_full_curve.py tests/test_hip_configs.py > $O/C.log 2>&1
cat $O/summary.txt; cat $O/B/rc.txt; cat $O/C/rc.txt
