#!/bin/bash
# two-level blocked Cholesky (+ explicit inverse) for 512 < n <= 2048: tests, timing against the 64-wide loop, BASELINE config 4 loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4dense; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_dense_gpu.py tests/test_harness_gpu.py tests/test_mll_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
rm -f $O/potrf.txt
for t in 0 1; do
  echo "== WISKI_POTRF_TWO_LEVEL=$t" >> $O/potrf.txt
  WISKI_POTRF_TWO_LEVEL=$t timeout 300 python tools/bench_small_potrf.py 600 900 1000 1500 2048 2>&1 | grep -v amdgpu >> $O/potrf.txt
  WISKI_POTRF_TWO_LEVEL=$t timeout 600 python tools/c4_probe.py 150 2>&1 | tail -1 >> $O/potrf.txt
done
cat $O/potrf.txt
