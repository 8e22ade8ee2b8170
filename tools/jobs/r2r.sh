#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o hyper -- python $R/tools/profile_hyper.py > $O/hyper.log 2>&1
cp /tmp/prof_h/hyper_kernel_stats.csv $O/hyper_kernel_stats.csv
cd $R; python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2r/hyper_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6, "over 25 steps ->", tot/1e6/25, "ms/step")
for r in rows[:22]:
    print(f"{r['Name'].split('(')[0][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
grep -E "cumulative|update\b|evaluate|_hyper_step|mll|backward|pcg|solve" $O/hyper.log | head -20
