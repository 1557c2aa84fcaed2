#!/bin/bash
# Tile-count quantisation of the decode kernel (128 x 64 tiles, 3 workgroups per CU = 768 slots): HIP-event time of the decode slot against the number of
# tiles, by sweeping F at B = 800 (7 row tiles) and B at F = 10000 (157 column tiles).  Run on the GPU box; output -> gpurun_out/r05_decode_quant.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out; OUT=gpurun_out/r05_decode_quant.txt; : > $OUT
run() {   # features batch
    local tiles=$(( (($2 + 127) / 128) * (($1 + 63) / 64) ))
    local line=$(timeout 120 python tools/kprof.py --precision f16x2 --strategy none --features $1 --batch $2 --rows $((2 * $2)) --steps 30 2>/dev/null | grep -E "decode_loss|dh_gemm|dw_gemm" | awk '{printf "%s %s  ", $1, $2}')
    echo "F=$1 B=$2 decode tiles=$tiles (= $(python -c "print('%.2f' % ($tiles / 768))") rounds of 768)  $line" | tee -a $OUT
}
for F in 3520 6976 7040 8448 10000 10496 12032 14016 14080 20992; do run $F 800; done
for B in 384 512 640 768 896 1024; do run 10000 $B; done
