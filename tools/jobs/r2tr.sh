#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o h -- python $GRAFT_REPO_ROOT/tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 --k 64 --reps 20 > /tmp/b.log 2>&1
grep half /tmp/b.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/tr/h_kernel_stats.csv')))
for r in rows[:6]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
