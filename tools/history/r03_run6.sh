#!/bin/bash
set -u
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/miner_timeline.py --lib dae_rnn_news_recommendation_amd/libdae_mp4.so > $O/timeline.txt 2>&1
cat $O/timeline.txt | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "miners" > $O/tests_miner.log 2>&1
tail -3 $O/tests_miner.log | cut -c1-300
for o in "" "--opt miner_tile=0"; do
  timeout 200 python tools/kprof.py $o >> $O/kprof.txt 2>&1
done
grep -v amdgpu.ids $O/kprof.txt
