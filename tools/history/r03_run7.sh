#!/bin/bash
# round-3 GPU pass 7: whole GPU suite on the lane-grid miner
set -u
O=gpurun_out/r3l
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 > $O/tests.log 2>&1
tail -25 $O/tests.log | cut -c1-300
