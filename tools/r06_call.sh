#!/bin/bash
mkdir -p gpurun_out/r06c15
timeout 300 python tools/region_trace.py --steps 20 --warmup 5 > gpurun_out/r06c15/region_trace.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06c15/region_trace.txt
