#!/bin/bash
# multi-column preconditioner / variance path: parity tests, 64-variance latency, kernel mix under rocprofv3
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "spectral or pcg" 2>&1 | grep -E "passed|failed|Error|assert" | head
for w in 8 4 8 4; do echo "slab waves $w"; WISKI_SLAB_WAVES=$w python tools/var_probe.py 64 | tail -2; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o v -- python $GRAFT_REPO_ROOT/tools/var_probe.py 64 > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_medians.py /tmp/prof_v/v_kernel_trace.csv | head -8
