#!/bin/bash
# measurement set of round 6: bench JSON (road-like headline, in-kernel SpMV stamps, 30^4 fp64 line, ELL gather), rocprofv3 kernel trace + stats of the
# headline loop, PMC traffic passes, kernel stats of the ELL probe, the no-launcher N = 2 self-test, reference-step / BO-loop probes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final; mkdir -p $O
cd $R
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -4 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
cp /tmp/prof_b/bench_kernel_stats.csv $O/bench_kernel_stats.csv; cp /tmp/prof_b/bench_kernel_trace.csv $O/bench_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o ell -- python $R/tools/gather_ell_probe.py --quick --final > $O/ell_prof.log 2>&1
cp /tmp/prof_e/ell_kernel_stats.csv $O/ell_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_ef -o f -- python $R/tools/gather_ell_probe.py --quick --final > $O/pmc_ef.log 2>&1
cd $R
python tools/trace_medians.py $O/bench_kernel_trace.csv > $O/kernel_medians.txt; head -14 $O/kernel_medians.txt
python tools/gap_report.py $O/bench_kernel_trace.csv > $O/gap_report.txt
python tools/spmv_trace_split.py $O/bench_kernel_trace.csv > $O/spmv_split.txt 2>&1; tail -3 $O/spmv_split.txt
python tools/spmv_overlap.py $O/bench_kernel_trace.csv > $O/spmv_overlap.txt 2>&1; head -8 $O/spmv_overlap.txt
python tools/pmc_traffic.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $O/pmc_traffic.json | grep -i "spmv\|scatter\|slab"
python tools/pmc_traffic.py /tmp/pmc_ef/f_counter_collection.csv /tmp/pmc_ef/f_counter_collection.csv $O/pmc_traffic_ell.json | grep -i "ell\|pack"
rm -f $O/bench_kernel_trace.csv
( WISKI_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 ) > $O/n2.log 2> $O/n2.err; tail -1 $O/n2.log > $O/n2.json
timeout 200 python tools/refstep_probe.py > $O/refstep_probe.txt 2>&1; tail -2 $O/refstep_probe.txt
timeout 600 python tools/c4_probe.py 200 2>&1 | tail -2 > $O/c4.txt; cat $O/c4.txt
python -c "
import json
r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], {k: r['roofline'].get(k) for k in ('frac','avg_launch_us','median_launch_us','frac_at_median_launch','launches_over_1.25x_median','event_frac','event_avg_launch_us','launches','traffic')})
print(r['roofline']['committed_take'])
for s in r.get('roofline_secondary', []): print(s['kernel'][:70], round(s['frac'],4), round(s['avg_launch_us'],1), s.get('plain_form_frac'))
print({k: e[k] for k in e if k.startswith('variance_ms') or k.startswith('reference_step_ms') or 'uniform' in k or 'errors' in k})
print(e['dense_regime']['reference_step'])
n=json.load(open('$O/n2.json')); print(n['n_gpus'], n['value'], n['collective'], n['config']['parallelism'][:80]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in n['extra'].items() if 'exchange' in k or 'error' in k})
"
