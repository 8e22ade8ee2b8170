#!/bin/bash
# per-XCD start offsets of the SpMV's waves with and without the dispatch events (is the 1.2 us skew of XCDs 0, 3-7 an artefact of the events?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6t; mkdir -p $O; rm -f $O/*.txt
cd $R
for n in 1000000 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --blocks 10 --sample-every $n --stamp-dump $O/st.npz > $O/b.log 2>&1
  echo "==== --sample-every $n" >> $O/report.txt
  python tools/stamp_report.py $O/st.npz 2>&1 | sed -n 1,24p >> $O/report.txt
  grep -o '"value": [0-9.]*' $O/b.log | head -1 >> $O/report.txt
  rm -f $O/st.npz
done
cat $O/report.txt
