#!/bin/bash
# light part of the LDS-DMA SpMV on paired row blocks (WISKI_SYM_PAIR3=1: 1 712 resident waves instead of 1 956): parity, back-to-back, bench A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6q; mkdir -p $O; rm -f $O/*.txt
cd $R
export PYTHONPATH=$R
WISKI_SYM_PAIR3=2 timeout 900 python -m pytest tests/test_hip_ops.py tests/test_two_level_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for p in 0 1 2 0 2; do
  echo "== probe WISKI_SYM_PAIR3=$p" >> $O/probe.txt
  WISKI_SYM_PAIR3=$p timeout 300 python tools/spmv_probe.py --reps 200 2>&1 | tail -4 >> $O/probe.txt
done
run() {
  echo "== $1" >> $O/ab.txt
  env $1 timeout 600 python bench.py --no-cpu-baseline --no-extras --blocks 10 > $O/b.log 2>&1
  python - $O/b.log >> $O/ab.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print(j["value"], j["ms_per_step"], r["avg_launch_us"], r["median_launch_us"], r["launches_over_1.25x_median"], r["launches"], round(r["frac"], 4), r["launch_us_percentiles_10_25_50_75_90_95_99"])
PY
}
for rep in 1 2 3; do
  run "WISKI_SYM_PAIR3=0"
  run "WISKI_SYM_PAIR3=1"
  run "WISKI_SYM_PAIR3=2"
done
cat $O/probe.txt $O/ab.txt
