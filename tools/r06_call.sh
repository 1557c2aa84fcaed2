#!/bin/bash
# round 6, final: the whole profile set at HEAD (per-kernel events = dispatch-stamped pairs, dae_plan_profile mode 3)
rm -rf gpurun_out/r06
bash tools/make_profile_report.sh r06 > gpurun_out/r06_report.log 2>&1
tail -5 gpurun_out/r06_report.log
