#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
WISKI_SYM_DMA=0 timeout 300 python tools/spmv_probe.py --reps 300 2>&1 | grep "half:" | sed "s/^/lds-window kernel /"
WISKI_SYM_DMA=0 timeout 300 python tools/spmv_probe.py --reps 100 --k 4 2>&1 | grep "half:" | sed "s/^/lds-window kernel k=4 /"
timeout 300 python tools/spmv_probe.py --reps 50 --dim 4 --grid 30 --dtype f64 --n 9568 2>&1 | grep "half:" | sed "s/^/30^4 f64 /"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1200
