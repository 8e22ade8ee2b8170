#!/bin/bash
# XCD-contiguous row blocks for the LDS-window half-stencil SpMV (fp64 / d != 3): 30^4 fp64 (C2), 50^3 fp64, 30^4 fp32
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4xcd2; mkdir -p $O
cd $R
rm -f $O/out.txt
for cfg in "4 30 f64 9568" "3 50 f64 100000" "4 30 f32 9568" "2 400 f64 100000"; do
  set -- $cfg
  for x in 0 1; do
    echo "== d=$1 g=$2 $3  WISKI_SYM_XCD=$x" >> $O/out.txt
    WISKI_SYM_XCD=$x timeout 600 python tools/spmv_probe.py --reps 30 --dim $1 --grid $2 --dtype $3 --n $4 2>&1 | grep "half\|diff" >> $O/out.txt
  done
done
cat $O/out.txt
