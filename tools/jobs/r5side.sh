#!/bin/bash
# A/B of the side stream's priority (lazy/two_level.py): SpMV dispatches of a bench trace beside / not beside a refresh kernel (tools/spmv_trace_split.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5side; mkdir -p $O; rm -f $O/out.txt
cd /tmp
for p in normal low normal low; do
  rm -rf /tmp/prof_b
  WISKI_TL_SIDE_PRIORITY=$p timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
  echo "== side stream priority $p" >> $O/out.txt
  python $R/tools/spmv_trace_split.py /tmp/prof_b/bench_kernel_trace.csv | grep "beside\|^all" >> $O/out.txt
  grep -o '"value": [0-9.]*' $O/prof.log | head -1 >> $O/out.txt
done
cat $O/out.txt
