#!/bin/bash
# gpurun with retries while no box / slot is free (exit code 3 or "transient"): tools/gpurun_retry.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out" | tail -${TAIL:-60}; exit $rc
done
echo "gpurun_retry: no slot after 40 tries"; exit 3
