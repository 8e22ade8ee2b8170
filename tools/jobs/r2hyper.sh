#!/bin/bash
# the reference-fidelity step at q = 1 (evaluate mean + variance -> Adam step on the MLL -> condition): wall per step, kernel
# mix and the kernel timeline of the last step under rocprofv3
python tools/hyper_probe.py 20 2>&1 | grep "_hyper_step"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o h -- python $GRAFT_REPO_ROOT/tools/profile_hyper.py > /tmp/b.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/tr/h_kernel_stats.csv')))
for r in rows[:18]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e3:7.1f} us")
print("sum all kernels ms:", sum(float(r['TotalDurationNs']) for r in rows) / 1e6)
PY
python $GRAFT_REPO_ROOT/tools/call_timeline.py /tmp/tr/h_kernel_trace.csv 400 | tail -150 > $GRAFT_REPO_ROOT/gpurun_out/hyper_timeline.txt
tail -3 $GRAFT_REPO_ROOT/gpurun_out/hyper_timeline.txt
grep -E "cumtime|_hyper_step|forward|backward|evaluate|predict" /tmp/b.log | head -12
