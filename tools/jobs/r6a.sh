#!/bin/bash
# round 6, first GPU pass: full GPU suite, the LDS-DMA ELL gather against the register-staged one, bench line with in-kernel stamps,
# the no-launcher N = 2 self-test of bench.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python tools/gather_ell_probe.py 2>&1 | grep -v amdgpu > $O/ell.txt; head -40 $O/ell.txt
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -3 $O/bench.err
( time WISKI_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-extras ) > $O/n2.log 2> $O/n2.err; tail -1 $O/n2.log > $O/n2.json; tail -5 $O/n2.err
python -c "
import json
r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], {k: r['roofline'][k] for k in ('frac','avg_launch_us','event_frac','event_avg_launch_us','launches','event_launches')})
for s in r.get('roofline_secondary', []): print(s['kernel'][:60], s['frac'], s['avg_launch_us'])
print(e.get('errors'))
r=json.load(open('$O/n2.json')); print(r['n_gpus'], r['value'], r['collective'], r['config']['parallelism'][:60])
"
