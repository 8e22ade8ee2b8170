#!/bin/bash
# XCD-contiguous row-block mapping of the LDS-DMA SpMV: knob sweep under the new mapping (probe, back to back)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4xcd; mkdir -p $O
cd $R
rm -f $O/sweep.txt
export WISKI_SYM_DMA_XCD=1
for cfg in "4 12 2" "4 0 2" "4 6 2" "4 9 2" "4 15 2" "4 18 2" "5 12 2" "6 12 2" "5 0 2" "6 0 2" "4 12 3" "4 12 2"; do
  set -- $cfg
  echo -n "parts=$1 delay=$2 nst=$3 : " >> $O/sweep.txt
  WISKI_SYM_DMA_PARTS=$1 WISKI_SYM_DMA_DELAY=$2 WISKI_SYM_DMA_NST=$3 timeout 300 python tools/spmv_probe.py --k 1 --reps 200 2>&1 | grep "half" >> $O/sweep.txt
done
cat $O/sweep.txt
