#!/bin/bash
# do the timed blocks of the default bench hit outliers (blocks over 1.5x the median)?  5 default-length runs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6r; mkdir -p $O; rm -f $O/*.txt
cd $R
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/b.log 2>&1
  python - $O/b.log >> $O/out.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]; e = j["extra"]
        print(j["value"], e.get("blocks"), e.get("blocks_over_1.5x_median_ms"), e.get("updates_per_s_median_block"), round(r["frac"], 4), r["launches"], e.get("block_ms_first_median_last_min"))
PY
done
cat $O/out.txt
