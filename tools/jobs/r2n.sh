#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2n/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'])
for k,v in r['extra'].items():
    if k!='dense_regime': print(k, v)
PY
