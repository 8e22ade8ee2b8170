#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_mll_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "^E  |passed|failed|^>|Error" $O/pytest.log | cut -c1-300 | head
for k in 16 64; do
WISKI_SPMM_COLS=0 timeout 300 python tools/spmv_probe.py --reps 30 --k $k 2>&1 | grep "half:" | sed "s/^/old k=$k /"
WISKI_SPMM_COLS=1 timeout 300 python tools/spmv_probe.py --reps 30 --k $k 2>&1 | grep "half:\|diff" | sed "s/^/cols k=$k /"
done
