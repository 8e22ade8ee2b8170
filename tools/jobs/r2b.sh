#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
WISKI_HIP_SO=$PWD/build/libwiski_dmatiming.so timeout 300 python tools/dma_timing.py 50 > $O/dma_timing.log 2>&1
cat $O/dma_timing.log
cd /tmp && WISKI_SYM_DMA=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o probe -- python $GRAFT_REPO_ROOT/tools/spmv_probe.py --reps 200 > /tmp/prof.log 2>&1
tail -3 /tmp/prof.log; find /tmp/prof_a | head -20
