#!/bin/bash
# round 6: cost of a cross-stream dependency by mechanism (event, stream memory ops, device-side flag)
mkdir -p gpurun_out/r06c25
hipcc --offload-arch=gfx950 -O2 tools/hop_probe.hip -o /tmp/hop_probe
(echo "# flag in uncached device memory"; timeout 30 /tmp/hop_probe; echo "rc $?"; echo "# flag in fine-grained device memory"; HOP_FINE=1 timeout 30 /tmp/hop_probe | grep -A1 "^S"; echo "rc $?") > gpurun_out/r06c25/hop_probe.txt 2>&1
cat gpurun_out/r06c25/hop_probe.txt
