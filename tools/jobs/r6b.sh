#!/bin/bash
# round 6: ELL gather tuning grid (tile size x wave count x tile order) and the cost of the SpMV's in-kernel stamps (A/B)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
timeout 900 python tools/gather_ell_probe.py --quick 2>&1 | grep -v amdgpu > $O/ell.txt
python - <<PY
import re
best={}
hdr=None
for ln in open('$O/ell.txt'):
    if ln.startswith('d='): hdr=ln.strip(); continue
    m=re.match(r'\s+(.*?)\s+([\d.]+) us\s+([\d.]+) TB/s', ln)
    if m and hdr: best.setdefault(hdr,[]).append((float(m.group(2)), m.group(1)))
for h,v in best.items():
    print(h); [print('   %8.1f us  %s'%x) for x in sorted(v)[:8]]; print('   ... registers:', [x for x in v if 'registers' in x[1]])
PY
for i in 1 2; do
  for ns in 0 1; do
    if [ $ns = 1 ]; then export WISKI_PROF_NOSTAMP=1; else unset WISKI_PROF_NOSTAMP; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_ns${ns}_$i.json
    python -c "
import json; r=json.load(open('$O/bench_ns${ns}_$i.json')); f=r['roofline']
print('nostamp=$ns', r['value'], {k: f.get(k) for k in ('avg_launch_us','median_launch_us','launches_over_1.25x_median','event_avg_launch_us','launches')})"
  done
done
