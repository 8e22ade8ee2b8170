#!/bin/bash
for i in 1 2; do
python bench.py --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['extra']['block_ms_first_median_last_min'], [ (x['kernel'][:12], round(x['avg_launch_us'],1)) for x in d['roofline_secondary']], {k: round(v,3) for k, v in d['extra'].items() if 'max_block' in k or k in ('step_ms_q1','step_ms_q64','variance_ms_per_64_queries_tol3e-3')})
"
done
