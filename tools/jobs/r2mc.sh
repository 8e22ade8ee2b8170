#!/bin/bash
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "spectral or pcg" 2>&1 | grep -E "passed|failed|Error|assert" | head
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed"
python tools/var_probe.py 64
WISKI_SLAB_MC_OFF=1 python tools/var_probe.py 64
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o v -- python $GRAFT_REPO_ROOT/tools/var_probe.py 64 > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_medians.py /tmp/prof_v/v_kernel_trace.csv | head -12
