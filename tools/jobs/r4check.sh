#!/bin/bash
# full GPU suite + the bench line (with extras)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4check; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -4 $O/bench.err
python -c "
import json; r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['net_of_empty_dispatch_frac'])
print(json.dumps(r['roofline_secondary'][0])[:900])
print({k: e[k] for k in e if k.startswith('variance_ms') or k.startswith('reference_step_ms') or 'uniform' in k})"
