#!/bin/bash
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | head
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export WISKI_NO_SPECULATION=1; echo "no speculation:"; else unset WISKI_NO_SPECULATION; echo "speculation:"; fi
  python bench.py --no-cpu-baseline --no-extras --blocks 40 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['extra']['cg_iters_per_step_mean'])
"; done
unset WISKI_NO_SPECULATION
python tools/q1_probe.py 2>&1 | tail -6
WISKI_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --blocks 3 --no-extras 2>&1 | grep -E "^\{|Error|error" | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --blocks 8 > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/gap_report.py /tmp/prof_b/bench_kernel_trace.csv | head -12
