#!/bin/bash
# new bench.py end to end + rocprofv3 kernel trace + PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -3 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
cp /tmp/prof_b/bench_kernel_stats.csv $O/bench_kernel_stats.csv; cp /tmp/prof_b/bench_kernel_trace.csv $O/bench_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_w.log 2>&1
cd $R
python tools/trace_medians.py $O/bench_kernel_trace.csv > $O/kernel_medians.txt; head -12 $O/kernel_medians.txt
python tools/pmc_traffic.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $O/pmc_traffic.json | head -12
