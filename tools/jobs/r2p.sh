#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "stream_step" 2>&1 | grep -E "passed|failed|^E |^>" | cut -c1-300 | head -12
timeout 900 python bench.py --no-cpu-baseline --no-extras > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2p/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'], r['extra'])
PY
