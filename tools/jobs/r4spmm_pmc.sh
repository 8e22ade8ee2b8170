#!/bin/bash
# instruction and cache counters of the 64-column half-stencil product, scalar-path kernel vs broadcast kernel (separate --pmc passes, kernel trace only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4spmm_pmc; mkdir -p $O; rm -f $O/pmc.txt
cd /tmp
for b in 0 1; do
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
  rm -rf /tmp/pm
  WISKI_SPMM_BCAST=$b timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm -o p -- python $R/tools/spmv_probe.py --k 64 --reps 6 > $O/log_$C.txt 2>&1
  if [ -f /tmp/pm/p_counter_collection.csv ]; then
    python - "$C" "$b" /tmp/pm/p_counter_collection.csv >> $O/pmc.txt <<'PY'
import csv, sys, statistics, collections
c, b, f = sys.argv[1:4]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c and "k_spmm_sym" in r["Kernel_Name"]:
        vals[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    print(f"WISKI_SPMM_BCAST={b} {c:22s} {k:48s} n={len(v):3d} median per launch {statistics.median(v):14.0f}")
PY
  else
    echo "WISKI_SPMM_BCAST=$b $C: not collected ($(tail -1 $O/log_$C.txt | cut -c1-120))" >> $O/pmc.txt
  fi
done; done
cat $O/pmc.txt
