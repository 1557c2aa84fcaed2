#!/bin/bash
# round 6, final: the whole profile set at HEAD (per-kernel events = dispatch-stamped pairs, dae_plan_profile mode 3) + the profile-mode test
rm -rf gpurun_out/r06
timeout 300 python -m pytest tests/test_hip_profile.py tests/test_hip_step.py -x -q -m gpu 2>&1 | tail -3
bash tools/make_profile_report.sh r06 > gpurun_out/r06_report.log 2>&1
tail -2 gpurun_out/r06_report.log
