#!/bin/bash
# the 64-column half-stencil product: scalar-path kernel (WISKI_SPMM_BCAST=0) against the DPP-broadcast one, tests + probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4spmm; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spmm_broadcast or spmv_and_kron or pcg_many_columns" 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
for b in ${SPMM_VARIANTS:-0 1}; do
  echo "== WISKI_SPMM_BCAST=$b" >> $O/probe.txt
  WISKI_SPMM_BCAST=$b timeout 300 python tools/spmv_probe.py --k 64 --reps 20 2>&1 | grep -v amdgpu.ids >> $O/probe.txt
done
cat $O/probe.txt
