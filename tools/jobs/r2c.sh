#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "spmv or pcg" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
WISKI_SYM_DMA_PARTS=7 timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "spmv or pcg" > $O/pytest7.log 2>&1; tail -2 $O/pytest7.log
for v in "0 4" "1 4" "1 7"; do
  set -- $v
  WISKI_SYM_DMA=$1 WISKI_SYM_DMA_PARTS=$2 timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep half | sed "s/^/dma=$1 parts=$2 /"
done
for p in 4 7; do
WISKI_SYM_DMA_PARTS=$p WISKI_HIP_SO=$PWD/build/libwiski_dmatiming.so timeout 300 python tools/dma_timing.py 50 > $O/dma_timing_p$p.log 2>&1
grep -v "alive\|amdgpu" $O/dma_timing_p$p.log
done
