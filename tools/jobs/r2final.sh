#!/bin/bash
# final measurement set of the round: bench JSON, rocprofv3 kernel trace + stats, PMC traffic passes, timelines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2final; mkdir -p $O
cd $R
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -4 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
cp /tmp/prof_b/bench_kernel_stats.csv $O/bench_kernel_stats.csv; cp /tmp/prof_b/bench_kernel_trace.csv $O/bench_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_w.log 2>&1
cd $R
python tools/trace_medians.py $O/bench_kernel_trace.csv > $O/kernel_medians.txt; head -14 $O/kernel_medians.txt
python tools/gap_report.py $O/bench_kernel_trace.csv > $O/gap_report.txt; cat $O/gap_report.txt
python tools/pmc_traffic.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $O/pmc_traffic.json | grep -i "spmv\|scatter"
WISKI_HIP_SO=$PWD/build/libwiski_dmatiming.so timeout 300 python tools/dma_timing.py 50 > $O/dma_wave_timeline.txt 2>&1
timeout 300 python tools/spmv_probe.py --reps 50 --dim 4 --grid 30 --dtype f64 --n 9568 > $O/spmv_30pow4_f64.txt 2>&1; grep half: $O/spmv_30pow4_f64.txt
timeout 120 ./build/stream_ubench > $O/stream_ubench.txt 2>&1; head -8 $O/stream_ubench.txt
