#!/bin/bash
# balanced XCD row ranges of the LDS-DMA SpMV: parity, then the all-dispatch statistic (committed form: mean 18.1-18.2, median 16.7-16.8, 1.90-1.92e7 at --blocks 10)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6y; mkdir -p $O; rm -f $O/*.txt
cd $R

for rep in 1 2 3 4; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --blocks 10 > $O/b.log 2>&1
  python - $O/b.log >> $O/out.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]; pc = r["launch_us_percentiles_10_25_50_75_90_95_99"]
        print(f"{j['value']:.4e} mean {r['avg_launch_us']:.2f} p10 {pc[0]:.2f} p25 {pc[1]:.2f} p50 {pc[2]:.2f} p75 {pc[3]:.2f} p90 {pc[4]:.2f} p99 {pc[6]:.2f} slow {r['launches_over_1.25x_median']}/{r['launches']} frac {r['frac']:.4f}")
PY
done

cat $O/out.txt
