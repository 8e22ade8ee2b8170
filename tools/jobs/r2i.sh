#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2i; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_prof.log 2>&1
find /tmp/prof_b -name "*.csv" | head
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) $O/bench_kernel_trace.csv
cd $GRAFT_REPO_ROOT; python tools/trace_medians.py $O/bench_kernel_trace.csv | head -14
