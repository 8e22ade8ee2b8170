#!/bin/bash
# does the lower column threshold of the broadcast SpMM (16 / 24 instead of 32) change the reference-fidelity step?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4thresh; mkdir -p $O
cd $R
for mn in 32 0; do
  echo "== WISKI_SPMM_COLS_MIN=$mn" >> $O/out.txt
  WISKI_SPMM_COLS_MIN=$mn timeout 300 python tools/refstep_probe.py 2>&1 | tail -3 >> $O/out.txt
  WISKI_SPMM_COLS_MIN=$mn WISKI_NO_SPECTRAL=1 timeout 300 python tools/var_probe.py 64 2>&1 | tail -2 >> $O/out.txt
done
cat $O/out.txt
