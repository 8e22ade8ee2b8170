#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
cp /tmp/prof_b/bench_kernel_trace.csv $O/bench_kernel_trace.csv
cd $R; python tools/gap_report.py $O/bench_kernel_trace.csv; tail -1 $O/prof.log | cut -c1-400
