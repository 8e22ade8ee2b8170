#!/bin/bash
# the many-column product at BASELINE config 2's size (30^4 fp64, A_h = 7.8 GB streams from HBM): scalar-path kernel vs broadcast kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c2spmm; mkdir -p $O; cd $R; rm -f $O/out.txt
for k in 32 64; do for b in 0 1; do
  echo "== 30^4 f64 k=$k WISKI_SPMM_BCAST=$b" >> $O/out.txt
  WISKI_SPMM_BCAST=$b timeout 600 python tools/spmv_probe.py --dim 4 --grid 30 --dtype f64 --n 9568 --k $k --reps 5 2>&1 | grep "half\|diff" >> $O/out.txt
done; done
cat $O/out.txt
