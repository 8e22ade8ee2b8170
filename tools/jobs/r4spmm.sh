#!/bin/bash
# the 64-column half-stencil product: scalar-path kernel (WISKI_SPMM_BCAST=0) against the DPP-broadcast one, tests + probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4spmm; mkdir -p $O
cd $R
test -n "$SPMM_SKIP_TESTS" || timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spmm_broadcast or spmv_and_kron or pcg_many_columns" 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
rm -f $O/probe.txt
for dt in f32 f64; do
for b in 0 1; do
for k in ${SPMM_KS:-64}; do
  echo "== $dt k=$k WISKI_SPMM_BCAST=$b" >> $O/probe.txt
  WISKI_SPMM_BCAST=$b timeout 300 python tools/spmv_probe.py --k $k --reps 20 --dtype $dt 2>&1 | grep "half\|diff" >> $O/probe.txt
done; done; done
cat $O/probe.txt
