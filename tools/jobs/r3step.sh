#!/bin/bash
# kernel timeline of one reference-fidelity step (profiles/r03_refstep_timeline.txt) + the small-matrix Cholesky timings
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3step; mkdir -p $O
cd /tmp
rm -rf /tmp/tr; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $R/tools/refstep_trace.py > $O/trace.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
test -n "$f" && python $R/tools/trace_timeline.py "$f" > $O/refstep_timeline.txt && head -3 $O/refstep_timeline.txt
cd $R
timeout 200 python tools/bench_small_potrf.py > $O/small_potrf.txt 2>&1; tail -10 $O/small_potrf.txt
timeout 300 python tools/refstep_soak.py > $O/refstep_soak.txt 2>&1; tail -9 $O/refstep_soak.txt
