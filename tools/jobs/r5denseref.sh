#!/bin/bash
# tools/dense_refstep_probe.py (reference-fidelity step in the dense regime), launch counts of each leg under rocprofv3 --kernel-trace --stats,
# and the kernel timeline of one step at 10^3 (RBF: truncated rank; Matern-5/2: full rank) and at 64 nodes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5denseref; mkdir -p $O; cd $R
timeout 900 python tools/dense_refstep_probe.py 2>&1 | grep -v amdgpu | tee $O/out.txt
for D in 1 2 3; do
cd /tmp; rm -rf /tmp/dr; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dr -o d -- python $R/tools/dense_refstep_probe.py $D > /dev/null 2>&1
D=$D python - <<'PY' | tee -a $O/out.txt
import csv, os
rows=list(csv.DictReader(open('/tmp/dr/d_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
legs = 1 if os.environ['D'] == '1' else 2
print('d = %s only (%d steps): GPU kernel time total ms' % (os.environ['D'], 100 * legs), tot/1e6, 'launches', n, 'per step', n/(100.0 * legs))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:24]:
    print(r['Name'][:110].ljust(110), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:7.1f} us")
PY
done
for C in "1 64 rbf" "3 10 rbf" "3 10 matern52"; do
cd /tmp; rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $R/tools/dense_refstep_trace.py $C > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
echo "== timeline of one step: $C" | tee -a $O/out.txt
test -n "$f" && python $R/tools/trace_timeline.py "$f" >> $O/out.txt
done
tail -60 $O/out.txt
