#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for v in "2 4" "2 6" "2 7" "3 6" "3 7"; do
  set -- $v
  WISKI_SYM_DMA=1 WISKI_SYM_DMA_NST=$1 WISKI_SYM_DMA_PARTS=$2 timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep half: | sed "s/^/nst=$1 parts=$2 /"
done
done
