#!/bin/bash
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "spectral or pcg" 2>&1 | grep -E "passed|failed|Error|assert" | head
for w in 8 4 8 4; do
  echo "slab waves $w:"; WISKI_SLAB_WAVES=$w python bench.py --no-cpu-baseline --no-extras --blocks 40 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['extra']['cg_iters_per_step_mean'])
"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --blocks 8 > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_medians.py /tmp/prof_b/bench_kernel_trace.csv | head -7
