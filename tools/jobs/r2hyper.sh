#!/bin/bash
python tools/hyper_probe.py 20 2>&1 | grep "_hyper_step"
PYPROF=1 python tools/hyper_probe.py 20 2>&1 | grep -v amdgpu | head -70 > gpurun_out/hyper_pyprof.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o h -- python $GRAFT_REPO_ROOT/tools/hyper_probe.py 20 > /tmp/b.log 2>&1
tail -1 /tmp/b.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/tr/h_kernel_stats.csv')))
tot = 0
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e3:7.1f} us")
print("sum all kernels ms:", sum(float(r['TotalDurationNs']) for r in rows) / 1e6)
PY
