#!/bin/bash
# multi-column preconditioner / variance path: 64-variance latency, kernel mix and the timeline of one call under rocprofv3
python tools/var_probe.py 64 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o v -- python $GRAFT_REPO_ROOT/tools/var_probe.py 64 > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_medians.py /tmp/prof_v/v_kernel_trace.csv | head -8
python $GRAFT_REPO_ROOT/tools/call_timeline.py /tmp/prof_v/v_kernel_trace.csv 1000
