#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c4; mkdir -p $O
cd $R
timeout 600 python tools/c4_probe.py 200 2>&1 | tail -2 > $O/c4.txt
cd /tmp; rm -rf /tmp/c4p
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4p -o c -- python $R/tools/c4_probe.py 120 > $O/prof.log 2>&1
cp /tmp/c4p/c_kernel_stats.csv $O/c4_kernel_stats.csv
tail -1 $O/prof.log >> $O/c4.txt
cat $O/c4.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/c4p/c_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('GPU kernel time total ms', tot/1e6)
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
