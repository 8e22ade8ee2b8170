#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for p in 4 6; do
  unset WISKI_HIP_SO
  WISKI_SYM_DMA=1 WISKI_SYM_DMA_PARTS=$p timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep "half:" | sed "s/^/base parts=$p /"
  export WISKI_HIP_SO=$PWD/build/libwiski_prio.so
  WISKI_SYM_DMA=1 WISKI_SYM_DMA_PARTS=$p timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep "half:" | sed "s/^/prio parts=$p /"
done
done
