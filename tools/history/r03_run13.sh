#!/bin/bash
set -u
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/w8_diag.py > $O/w8_diag.log 2>&1
cat $O/w8_diag.log | cut -c1-300
