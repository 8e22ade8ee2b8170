python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --blocks 60 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['extra']; print(d['value'], d['ms_per_step'], e['updates_per_s_median_block'], e['updates_per_s_all_blocks'], e['block_ms_first_median_last_min'], e['blocks_dropped'], e['cg_iters_per_step_mean'])"; done
python bench.py --no-cpu-baseline --no-extras --blocks 40 --stream clustered 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['extra']; print('clustered', d['value'], d['ms_per_step'], e['cg_iters_per_step_mean'])"
