#!/bin/bash
# tools/dense_refstep_probe.py (reference-fidelity step in the dense regime) + launch counts of each leg under rocprofv3 --kernel-trace --stats.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5denseref; mkdir -p $O; cd $R
timeout 900 python tools/dense_refstep_probe.py 2>&1 | grep -v amdgpu | tee $O/out.txt
for D in 1 2 3; do
cd /tmp; rm -rf /tmp/dr; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dr -o d -- python $R/tools/dense_refstep_probe.py $D > /dev/null 2>&1
D=$D python - <<'PY' | tee -a $O/out.txt
import csv, os
rows=list(csv.DictReader(open('/tmp/dr/d_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
print('d = %s only (100 steps): GPU kernel time total ms' % os.environ['D'], tot/1e6, 'launches', n, 'per step', n/100)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
    print(r['Name'][:110].ljust(110), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:7.1f} us")
PY
done
