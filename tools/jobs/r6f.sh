#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 600 python tools/n2_iters_probe.py 2>&1 | grep -v amdgpu > $O/n2_iters.txt; cat $O/n2_iters.txt
