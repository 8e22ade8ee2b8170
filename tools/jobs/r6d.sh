#!/bin/bash
# stamps by plain per-wave stores: A/B against no stamps (events), then the full bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
for i in 1 2; do
  for ns in 0 1; do
    if [ $ns = 1 ]; then export WISKI_PROF_NOSTAMP=1; else unset WISKI_PROF_NOSTAMP; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_ns${ns}_$i.json
    python -c "
import json; r=json.load(open('$O/bench_ns${ns}_$i.json')); f=r['roofline']
print('nostamp=$ns', r['value'], {k: f.get(k) for k in ('frac','avg_launch_us','median_launch_us','launches_over_1.25x_median','event_avg_launch_us','launches')})"
  done
done
unset WISKI_PROF_NOSTAMP
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -3 $O/bench.err
python -c "
import json
r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], {k: r['roofline'].get(k) for k in ('frac','avg_launch_us','median_launch_us','frac_at_median_launch','event_frac','event_avg_launch_us','launches')})
for s in r.get('roofline_secondary', []): print(s['kernel'][:70], round(s['frac'],4), round(s['avg_launch_us'],1), s.get('plain_form_frac'))
print(e.get('errors'))
"
