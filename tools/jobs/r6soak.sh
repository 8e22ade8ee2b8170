#!/bin/bash
# soak runs re-taken with the round-6 library (switches pruned, stems rewritten): streaming path at 50^3, the reference step at 50^3 and on two small grids
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6soak; mkdir -p $O; cd $R
( timeout 600 python tools/soak.py 3 2>&1 | grep -v amdgpu | tail -6 ) > $O/soak.txt; cat $O/soak.txt
( timeout 900 python tools/refstep_soak.py 2>&1 | grep -v amdgpu | tail -12 ) > $O/refstep_soak.txt; tail -4 $O/refstep_soak.txt
( timeout 900 python tools/dense_refstep_soak.py 3 10 matern52 2>&1 | grep -v amdgpu | tail -10 ) > $O/dense_soak_10c_m52.txt; tail -3 $O/dense_soak_10c_m52.txt
( timeout 900 python tools/dense_refstep_soak.py 2 30 matern12 2>&1 | grep -v amdgpu | tail -10 ) > $O/dense_soak_30s_m12.txt; tail -3 $O/dense_soak_30s_m12.txt
