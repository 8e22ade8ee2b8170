#!/bin/bash
# measurement set of round 5: bench JSON (road-like headline), rocprofv3 kernel trace + stats of the headline loop, PMC traffic passes, the
# reference step at 50^3 (probe + timeline) and on the small grids (tools/jobs/r5denseref.sh), Cholesky sizes, BASELINE config-4 loop, N = 2 self-test
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final; mkdir -p $O
cd $R
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -4 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
cp /tmp/prof_b/bench_kernel_stats.csv $O/bench_kernel_stats.csv; cp /tmp/prof_b/bench_kernel_trace.csv $O/bench_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_w.log 2>&1
cd $R
python tools/trace_medians.py $O/bench_kernel_trace.csv > $O/kernel_medians.txt; head -14 $O/kernel_medians.txt
python tools/gap_report.py $O/bench_kernel_trace.csv > $O/gap_report.txt
python tools/spmv_trace_split.py $O/bench_kernel_trace.csv > $O/spmv_split.txt 2>&1; tail -3 $O/spmv_split.txt
python tools/pmc_traffic.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $O/pmc_traffic.json | grep -i "spmv\|scatter\|slab"
rm -f $O/bench_kernel_trace.csv
cd /tmp
rm -rf /tmp/tr; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $R/tools/refstep_trace.py > $O/trace.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
test -n "$f" && python $R/tools/trace_timeline.py "$f" > $O/refstep_timeline.txt && head -1 $O/refstep_timeline.txt
cd $R
timeout 200 python tools/bench_small_potrf.py 2>&1 | grep -v amdgpu > $O/small_potrf.txt; timeout 200 python tools/bench_small_potrf.py 600 900 1000 2>&1 | grep -v amdgpu >> $O/small_potrf.txt; grep "327\|480\|1000" $O/small_potrf.txt
timeout 200 python tools/refstep_probe.py > $O/refstep_probe.txt 2>&1; tail -2 $O/refstep_probe.txt
timeout 600 python tools/c4_probe.py 200 2>&1 | tail -2 > $O/c4.txt; cat $O/c4.txt
bash tools/jobs/r5denseref.sh > /dev/null 2>&1; cp $R/gpurun_out/r5denseref/out.txt $O/dense_refstep.txt; head -10 $O/dense_refstep.txt
timeout 300 python tools/tl_cold_probe.py 2>&1 | grep -v amdgpu > $O/tl_cold_probe.txt; grep "64 columns\|rebuild" $O/tl_cold_probe.txt
bash tools/jobs/r5n2.sh > $O/n2.txt 2>&1; cp $R/gpurun_out/r5n2/out.txt $O/n2_out.txt; tail -3 $O/n2.txt
python -c "
import json; r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['net_of_empty_dispatch_frac'])
print({k: e[k] for k in e if k.startswith('variance_ms') or k.startswith('reference_step_ms') or 'uniform' in k or 'errors' in k or 'plain_step' in k})"
