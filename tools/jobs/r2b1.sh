#!/bin/bash
# generic: GPU suite, then the headline A/B of an environment switch given as $1 (e.g. WISKI_FUSED_ITER) with values 1 / 0
V=${1:-WISKI_FUSED_ITER}
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | head -5
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['extra']; print('   ', d['value'], d['ms_per_step'], e['updates_per_s_median_block'], d['roofline']['avg_launch_us'], d['roofline']['frac'], e['cg_iters_per_step_mean'], e['blocks_dropped'])"; }
for o in 1 0 1 0; do echo "$V=$o"; env $V=$o bash -c "$(declare -f run); run --blocks 60"; done
echo clustered; for o in 1 0; do env $V=$o bash -c "$(declare -f run); run --blocks 40 --stream clustered"; done
