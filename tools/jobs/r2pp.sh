#!/bin/bash
python tools/policy_probe.py 20 | cut -c1-400
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | head
run() { python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['extra']; print('   ', d['value'], d['ms_per_step'], 'median-based', e['updates_per_s_median_block'], 'launches', d['roofline']['launches'], d['roofline']['avg_launch_us'], d['roofline']['frac'], e['cg_iters_per_step_mean'], e['blocks_dropped'])
"; }
run --blocks 40; run --blocks 40; run --blocks 40 --stream clustered
