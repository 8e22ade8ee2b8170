#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
for p in 5 6; do
WISKI_SYM_DMA_PARTS=$p timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "spmv" 2>&1 | tail -1
done
for p in 4 5 6 7; do
  WISKI_SYM_DMA=1 WISKI_SYM_DMA_PARTS=$p timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep half: | sed "s/^/parts=$p /"
done
for p in 4 5; do
WISKI_SYM_DMA_PARTS=$p WISKI_HIP_SO=$PWD/build/libwiski_dmatiming.so timeout 300 python tools/dma_timing.py 50 > $O/dma_timing_p$p.log 2>&1
grep -v "alive\|amdgpu" $O/dma_timing_p$p.log
done
