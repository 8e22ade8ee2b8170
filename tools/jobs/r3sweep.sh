#!/bin/bash
# in-bench sweep of the DMA SpMV's launch parameters (they were tuned back to back in tools/spmv_probe.py)
cd $GRAFT_REPO_ROOT
for P in 4 5 6; do for D in 0 6 12 18; do
  export WISKI_SYM_DMA_PARTS=$P WISKI_SYM_DMA_DELAY=$D
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts', os.environ['WISKI_SYM_DMA_PARTS'], 'delay', os.environ['WISKI_SYM_DMA_DELAY'], 'updates/s %.4g' % r['value'], 'ms %.4f' % r['ms_per_step'], 'spmv us %.2f' % r['roofline']['avg_launch_us'], 'frac %.3f' % r['roofline']['frac'])"
done; done
