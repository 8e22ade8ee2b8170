#!/bin/bash
# does a low-priority refresh stream (two-level block) leave the solver's kernels alone?  bench headline + per-dispatch SpMV time
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4prio; mkdir -p $O
cd $R
rm -f $O/out.txt
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())" >> $O/out.txt 2>&1
for p in default low default low; do
  echo "== WISKI_TL_SIDE_PRIORITY=$p" >> $O/out.txt
  WISKI_TL_SIDE_PRIORITY=$p timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_us'], r['roofline']['frac'], r['extra']['cg_iters_per_step_mean'])" >> $O/out.txt
done
cat $O/out.txt
