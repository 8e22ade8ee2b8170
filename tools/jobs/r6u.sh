#!/bin/bash
# bench.py with the dispatch events outside the timed region: three default runs + the no-launcher N = 2 self-test
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6u; mkdir -p $O; rm -f $O/*.txt
cd $R
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/b$i.log 2>&1
  python - $O/b$i.log >> $O/out.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]; e = j["extra"]
        print(j["value"], j["ms_per_step"], e.get("blocks"), e.get("blocks_over_1.5x_median_ms"), round(r["frac"], 4), r["avg_launch_us"], r["median_launch_us"], r["launches_over_1.25x_median"], r["launches"], r["event_launches"], r["event_avg_launch_us"], round(r["event_frac"], 4))
PY
done
( WISKI_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 ) > $O/n2.log 2> $O/n2.err; tail -1 $O/n2.log | python -c "
import json,sys
n=json.loads(sys.stdin.read()); print(n['n_gpus'], n['value'], n['roofline']['frac'], n['roofline']['launches'], n['roofline']['event_launches'], n['extra'].get('errors'))"
tail -3 $O/n2.err
cat $O/out.txt
