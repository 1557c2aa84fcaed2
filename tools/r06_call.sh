#!/bin/bash
# round 6: the committed profile set (one run on one box), right behind smoke()
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -6 gpurun_out/r06_smoke.txt
bash tools/make_profile_report.sh r06 > gpurun_out/r06_report.log 2>&1; tail -30 gpurun_out/r06_report.log
