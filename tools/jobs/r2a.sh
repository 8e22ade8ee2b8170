#!/bin/bash
# round-2 job A: correctness of the LDS-DMA SpMV + A/B timing against the LDS-window kernel
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q > $O/pytest_hip_ops.log 2>&1; echo "rc=$?" >> $O/pytest_hip_ops.log
tail -5 $O/pytest_hip_ops.log
for v in "0 2" "1 2" "1 3"; do
  set -- $v
  WISKI_SYM_DMA=$1 WISKI_SYM_DMA_NST=$2 timeout 300 python tools/spmv_probe.py --reps 200 > $O/probe_dma$1_nst$2.log 2>&1
  tail -4 $O/probe_dma$1_nst$2.log
done
# rocprof cross-check of the per-dispatch event timing
(cd /tmp && WISKI_SYM_DMA=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o probe -- python $GRAFT_REPO_ROOT/tools/spmv_probe.py --reps 200 > /dev/null 2>&1)
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/probe_kernel_stats.csv
head -8 $O/probe_kernel_stats.csv | cut -c1-200
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-1500
