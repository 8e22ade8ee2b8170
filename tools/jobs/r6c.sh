#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "ell" 2>&1 | tail -15 > $O/tests.log; tail -6 $O/tests.log
timeout 600 python tools/gather_ell_probe.py --quick --policy 2>&1 | grep -v amdgpu | grep "d=\|grid-aware\|dma P=8 waves=512 \|regis" > $O/ell.txt; cat $O/ell.txt
