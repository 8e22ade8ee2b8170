#!/bin/bash
# final measurement set of round 3: bench JSON, rocprofv3 kernel trace + stats of the headline loop, PMC traffic passes, and the
# kernel mixes of the variance and hyper-parameter-step paths (default = spectral factor, and the PCG path)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -4 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
cp /tmp/prof_b/bench_kernel_stats.csv $O/bench_kernel_stats.csv; cp /tmp/prof_b/bench_kernel_trace.csv $O/bench_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 2 > $O/pmc_w.log 2>&1
cd $R
python tools/trace_medians.py $O/bench_kernel_trace.csv > $O/kernel_medians.txt; head -12 $O/kernel_medians.txt
python tools/gap_report.py $O/bench_kernel_trace.csv > $O/gap_report.txt
python tools/pmc_traffic.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $O/pmc_traffic.json | grep -i "spmv\|scatter"
cd /tmp
for tag in "" _pcg; do
  if [ "$tag" = "_pcg" ]; then export WISKI_NO_SPECTRAL=1; else unset WISKI_NO_SPECTRAL; fi
  python $R/tools/var_probe.py 64 2>&1 | tail -1
  rm -rf /tmp/pv$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv$tag -o v -- python $R/tools/var_probe.py 64 > $O/var$tag.log 2>&1
  cp /tmp/pv$tag/v_kernel_stats.csv $O/variance${tag}_kernel_stats.csv
  python $R/tools/hyper_probe.py 20 2>&1 | tail -1
  rm -rf /tmp/ph$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph$tag -o h -- python $R/tools/hyper_probe.py 20 > $O/hyper$tag.log 2>&1
  cp /tmp/ph$tag/h_kernel_stats.csv $O/hyper${tag}_kernel_stats.csv
done
unset WISKI_NO_SPECTRAL
cd $R
WISKI_HIP_SO=$PWD/build/libwiski_dmatiming.so timeout 300 python tools/dma_timing.py 50 > $O/dma_wave_timeline.txt 2>&1 || true
python -c "
import json; r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['net_of_empty_dispatch_frac'])
print({k: e[k] for k in e if k.startswith('variance_ms') or k.startswith('reference_step_ms') or 'clustered' in k or 'errors' in k})"
