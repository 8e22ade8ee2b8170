#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4trace; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
python $R/tools/spmv_trace_split.py /tmp/prof_b/bench_kernel_trace.csv | tee $O/spmv_split.txt
