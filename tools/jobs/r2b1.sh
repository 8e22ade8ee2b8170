#!/bin/bash
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --blocks 40 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
"; done
python tools/var_probe.py 64 2>&1 | grep variance
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o h -- python $GRAFT_REPO_ROOT/tools/var_probe.py 64 > /tmp/b.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/tr/h_kernel_stats.csv')))
for r in rows[:10]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
