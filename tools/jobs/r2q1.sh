#!/bin/bash
python tools/q1_probe.py 1 400 2>&1 | grep "q ="
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/tools/q1_probe.py 1 100 > /tmp/b.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2
