#!/bin/bash
# round-3 evidence for the statistics scatter: the memory-side atomic transaction rate of the hardware (micro-benchmarks) and the
# scatter kernel's own memory-side request counters (separate rocprofv3 --pmc passes, kernel trace only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ev; mkdir -p $O
cd $R
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -o build/atomic_scope_ubench tools/ubench/atomic_scope_ubench.hip 2> $O/build.log
hipcc --offload-arch=gfx950 -O3 -o build/atomic_ubench tools/ubench/atomic_ubench.hip 2>> $O/build.log
( echo "== tools/ubench/atomic_scope_ubench.hip (86 MB target, runs of 8 floats at random 32-byte aligned places) =="; timeout 120 ./build/atomic_scope_ubench;
  echo; echo "== tools/ubench/atomic_ubench.hip =="; timeout 120 ./build/atomic_ubench ) > $O/atomic_ubench.txt 2>&1
( echo; echo "== tools/scatter_probe.py 4096 (kernel-only times of the absorb's parts, 50^3 fp32) =="; timeout 120 python tools/scatter_probe.py 4096 ) >> $O/atomic_ubench.txt 2>&1
cd /tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
grep -i -o -E "TCC_[A-Z0-9_]*(ATOMIC|WRREQ|WRITE|RDREQ|EA0_WR)[A-Z0-9_]*" $O/counters_avail.txt | sort -u > $O/tcc_counters.txt
for C in TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_WRITE_sum; do
  rm -rf /tmp/pm_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm_$C -o p -- python $R/tools/scatter_probe.py 4096 > $O/pmc_$C.log 2>&1
  if [ -f /tmp/pm_$C/p_counter_collection.csv ]; then
    python - "$C" /tmp/pm_$C/p_counter_collection.csv /tmp/pm_$C/p_kernel_trace.csv >> $O/pmc_scatter.txt <<'PY'
import csv, sys, statistics, collections
c, f, t = sys.argv[1:4]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c and "scatter" in r["Kernel_Name"]:
        vals[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    if "scatter" in r["Kernel_Name"]:
        dur[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in vals.items():
    v = sorted(v); d = sorted(dur.get(k, [0]))
    # the probe launches 6 variants x 21 times: report the largest group (all of b, cnt, res, A) = the top sixth
    top = v[-max(1, len(v) // 6):]
    print(f"{c:24s} {k:60s} n={len(v):4d} median {statistics.median(v):12.0f}  full-absorb launches (top sixth) median {statistics.median(top):12.0f}  kernel us (median of all) {statistics.median(d):7.1f} max {d[-1]:7.1f}")
PY
  else
    echo "$C: not collected ($(tail -1 $O/pmc_$C.log))" >> $O/pmc_scatter.txt
  fi
done
cat $O/atomic_ubench.txt | tail -40; cat $O/pmc_scatter.txt; head -30 $O/tcc_counters.txt
