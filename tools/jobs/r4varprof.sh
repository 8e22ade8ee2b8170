#!/bin/bash
# kernel mix of 64 predictive variances on the PCG path (WISKI_NO_SPECTRAL=1), uniform 50^3 stream
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4varprof; mkdir -p $O
cd /tmp
export WISKI_NO_SPECTRAL=1
rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o v -- python $R/tools/var_probe.py 64 > $O/var.log 2>&1
cp /tmp/pv/v_kernel_stats.csv $O/variance_pcg_kernel_stats.csv
head -12 $O/variance_pcg_kernel_stats.csv | cut -c1-90,150-400 | sed 's/([^"]*"/"/'
tail -3 $O/var.log
