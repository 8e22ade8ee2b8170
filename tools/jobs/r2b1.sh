#!/bin/bash
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --blocks 40 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
"; done
