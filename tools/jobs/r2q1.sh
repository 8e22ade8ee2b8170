#!/bin/bash
python bench.py --no-cpu-baseline --blocks 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step']); print({k: v for k, v in d['extra'].items() if 'step_ms' in k})
    elif 'Error' in l or 'Traceback' in l or 'line' in l: print(l)
"
