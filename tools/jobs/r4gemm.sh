#!/bin/bash
# the 32 x 32 / 64 x 64 tile switch of wiski_gemm at the dense regime's mid sizes; what it does to the factorisations and to the config-4 loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4gemm; mkdir -p $O; cd $R; rm -f $O/out.txt
for mx in 128 0; do
  echo "== WISKI_GEMM32_MAX_TILES=$mx (0: default = 500 fp64 / 300 fp32)" >> $O/out.txt
  WISKI_GEMM32_MAX_TILES=$mx timeout 300 python tools/gemm_mid_probe.py 2>&1 | grep -v amdgpu >> $O/out.txt
  WISKI_GEMM32_MAX_TILES=$mx timeout 300 python tools/bench_small_potrf.py 1000 2048 2>&1 | grep -v amdgpu >> $O/out.txt
  WISKI_GEMM32_MAX_TILES=$mx timeout 600 python tools/c4_probe.py 150 2>&1 | tail -1 >> $O/out.txt
done
cat $O/out.txt
timeout 900 python -m pytest tests/test_dense_gpu.py -x -q -m gpu 2>&1 | tail -2
