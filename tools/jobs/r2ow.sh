#!/bin/bash
# owner-computes absorb: full GPU suite, A/B of the headline (WISKI_OWNER_MIN_POINTS=0 keeps the atomic form)
python -m pytest tests -m gpu -q -x 2>&1 | tail -1
run() { python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['extra']; print('   ', d['value'], d['ms_per_step'], 'median-based', e['updates_per_s_median_block'], d['roofline']['avg_launch_us'], d['roofline']['frac'], e['cg_iters_per_step_mean'], e['blocks_dropped'])
"; }
for o in 4096 0 4096 0; do echo "owner min points $o"; WISKI_OWNER_MIN_POINTS=$o run --blocks 40; done
echo clustered; WISKI_OWNER_MIN_POINTS=4096 run --blocks 40 --stream clustered; WISKI_OWNER_MIN_POINTS=0 run --blocks 40 --stream clustered
