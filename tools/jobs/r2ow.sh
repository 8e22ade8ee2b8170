#!/bin/bash
# owner-computes absorb at q = 4096: does a non-temporal write-back keep the SpMV fast?
run() { python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['extra']; print('   ', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
"; }
export WISKI_OWNER_MIN_POINTS=4096
echo "plain stores"; run --blocks 40
echo "nt stores"; WISKI_HIP_SO=$PWD/build/libwiski_ntstores.so run --blocks 40
echo "atomic form"; WISKI_OWNER_MIN_POINTS=0 run --blocks 40
