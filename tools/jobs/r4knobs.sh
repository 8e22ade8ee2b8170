#!/bin/bash
# the documented A/B switches still select working code paths: the kernel / dense / model test files under each fallback
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4knobs; mkdir -p $O; cd $R; rm -f $O/out.txt
run() { echo "== $1" >> $O/out.txt; env $1 timeout 1500 python -m pytest $2 -x -q -m gpu 2>&1 | tail -1 >> $O/out.txt; }
run "WISKI_SPMM_BCAST=0" "tests/test_hip_ops.py tests/test_model_gpu.py"
run "WISKI_SYM_XCD=0" "tests/test_hip_ops.py tests/test_model_gpu.py tests/test_distributed_gpu.py"
run "WISKI_SYM_XCD=1" "tests/test_hip_ops.py"
run "WISKI_POTRF_TWO_LEVEL=0" "tests/test_dense_gpu.py tests/test_harness_gpu.py"
run "WISKI_GEMM32_MAX_TILES=128" "tests/test_dense_gpu.py tests/test_spectral_gpu.py"
run "WISKI_TL_QUIET_POTRF=0" "tests/test_two_level_gpu.py"
run "WISKI_POTRF_COOP=0" "tests/test_dense_gpu.py"
cat $O/out.txt
