#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -3 $O/bench.err
python -c "
import json
r=json.load(open('$O/bench.json')); f=r['roofline']
print(r['value'], r['ms_per_step'], {k: f.get(k) for k in ('frac','avg_launch_us','median_launch_us','launches_over_1.25x_median','event_frac','launches')}, f.get('launch_us_percentiles_10_25_50_75_90_95_99'))
for s in r.get('roofline_secondary', []): print(s['kernel'][:60], round(s['frac'],4), round(s['avg_launch_us'],1))
print(r['extra'].get('errors'), r.get('cpu_baseline',{}).get('value'))"
