#!/bin/bash
set -u
O=gpurun_out/r3t
mkdir -p $O
timeout 120 ./tools/lds_stream_ubench > $O/lds_stream_ubench.txt 2>&1
cat $O/lds_stream_ubench.txt
