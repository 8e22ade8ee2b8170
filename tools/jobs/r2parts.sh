#!/bin/bash
for i in 1 2 3; do python tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 --reps 100 2>&1 | grep -E "half:|diff"; done
python -m pytest tests/test_hip_ops.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
