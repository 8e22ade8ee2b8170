#!/bin/bash
export TMPDIR=/tmp
for a in 0 1 2 3 4; do
  if [ $a = 0 ]; then unset WISKI_HIP_SO; else export WISKI_HIP_SO=$PWD/build/libwiski_abl$a.so; fi
  WISKI_SYM_DMA=1 WISKI_SYM_DMA_PARTS=6 timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep "half:" | sed "s/^/ablate=$a /"
done
