#!/bin/bash
# final measurement set of round 4: bench JSON (road-like headline), rocprofv3 kernel trace + stats of the headline loop, PMC traffic
# passes, the knob sweep of the two-level preconditioner, and the kernel timeline of one reference-fidelity step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final; mkdir -p $O
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
python tools/pmc_traffic.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $O/pmc_traffic.json | grep -i "spmv\|scatter\|slab"
rm -f $O/bench_kernel_trace.csv
timeout 600 python tools/two_level_probe.py 50 192,2,1.2 192,2,1.1 256,2,1.2 128,2,1.2 96,2,1.2 > $O/two_level_probe.txt 2>&1; grep -v "its \[" $O/two_level_probe.txt | tail -14
cd /tmp
rm -rf /tmp/tr; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $R/tools/refstep_trace.py > $O/trace.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
test -n "$f" && python $R/tools/trace_timeline.py "$f" > $O/refstep_timeline.txt && head -1 $O/refstep_timeline.txt
cd $R
timeout 200 python tools/bench_small_potrf.py > $O/small_potrf.txt 2>&1; grep "327\|480" $O/small_potrf.txt
timeout 200 python tools/refstep_probe.py > $O/refstep_probe.txt 2>&1; tail -2 $O/refstep_probe.txt
# the many-column half-stencil product: scalar-path kernel (WISKI_SPMM_BCAST=0) against the DPP-broadcast one, both precisions
rm -f $O/spmm_bcast_probe.txt
for dt in f32 f64; do for b in 0 1; do for k in 16 24 32 64 100; do
  echo "== $dt k=$k WISKI_SPMM_BCAST=$b" >> $O/spmm_bcast_probe.txt
  WISKI_SPMM_BCAST=$b timeout 300 python tools/spmv_probe.py --k $k --reps 20 --dtype $dt 2>&1 | grep "half\|diff" >> $O/spmm_bcast_probe.txt
done; done; done
grep -A1 "k=64" $O/spmm_bcast_probe.txt | grep -v "^--"
# dense regime: GEMM / Cholesky / TRSM at n = 1024 .. 4096, the two-level factorisation at the model's sizes, BASELINE config 4 loop
timeout 300 python tools/bench_dense.py 2>&1 | grep -v amdgpu > $O/dense.txt; grep '"n": 1024' $O/dense.txt
timeout 300 python tools/bench_small_potrf.py 600 900 1000 1500 2048 2>&1 | grep -v amdgpu >> $O/dense.txt
timeout 600 python tools/c4_probe.py 200 2>&1 | tail -1 >> $O/dense.txt; tail -1 $O/dense.txt
# kernel mix of 64 predictive variances on the PCG path
cd /tmp
rm -rf /tmp/pv; WISKI_NO_SPECTRAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o v -- python $R/tools/var_probe.py 64 > $O/var_pcg.log 2>&1
cp /tmp/pv/v_kernel_stats.csv $O/variance_pcg_kernel_stats.csv; grep "variance of" $O/var_pcg.log | tail -1
cd $R
python -c "
import json; r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['net_of_empty_dispatch_frac'])
print({k: e[k] for k in e if k.startswith('variance_ms') or k.startswith('reference_step_ms') or 'uniform' in k or 'errors' in k or 'two_level' in k})"
