#!/bin/bash
# round 6: option A/Bs in the new default mode (f16x2h, three decode terms / four dW terms): wide decode tiles, row-major delta2 (no delta2^T image)
mkdir -p gpurun_out/r06c17
O=gpurun_out/r06c17
for o in "" "--opt decode_bn=128" "--opt dw_tr=1" "--opt decode_bn=128 --opt dw_tr=1" "--opt dw_pair=0"; do
  timeout 200 python tools/kprof.py --precision f16x2h $o 2>&1 | grep -E "^==|decode_loss|dw_gemm|encode_gemm" | sed 's/info=.*//' >> $O/ab.txt
done
cat $O/ab.txt
