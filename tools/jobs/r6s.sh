#!/bin/bash
# the driver's line once more on the final tree (bench.py only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s; mkdir -p $O
cd $R
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -4 $O/bench.err
python -c "
import json
r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], e.get('blocks_over_1.5x_median_ms'), {k: r['roofline'].get(k) for k in ('frac','avg_launch_us','median_launch_us','launches_over_1.25x_median','launches','event_frac')})
for s in r.get('roofline_secondary', []): print(s['kernel'][:60], round(s['frac'],4), round(s['avg_launch_us'],1), s.get('kernel_only_us'), s.get('kernel_only_frac'))
print(e.get('errors'), r['cpu_baseline']['value'])"
