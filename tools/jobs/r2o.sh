#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head -5
# N > 1 code path of bench.py on one device (gloo self-test hook; never used for reported numbers)
WISKI_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --blocks 3 > $O/bench2.log 2> $O/bench2.err; tail -1 $O/bench2.log | cut -c1-1800; tail -3 $O/bench2.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2o/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'])
for k,v in r['extra'].items():
    if k!='dense_regime': print(k, v)
PY
