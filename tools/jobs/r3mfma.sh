#!/bin/bash
# MFMA activity of the dense GEMM (both tile sizes) by PMC: counter-only pass + an un-instrumented pass for the durations
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3mfma; mkdir -p $O
cd /tmp
python $R/tools/bench_dense.py > $O/bench_dense.txt 2>&1
rm -rf /tmp/mf /tmp/mt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/mf -o p -- python $R/tools/bench_dense.py > $O/pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt -o t -- python $R/tools/bench_dense.py > $O/trace.log 2>&1
python - /tmp/mf/p_counter_collection.csv /tmp/mt/t_kernel_trace.csv > $O/mfma_pmc.txt <<'PY'
import csv, sys, collections, statistics
cf, tf = sys.argv[1:3]
cnt = collections.defaultdict(list)
for r in csv.DictReader(open(cf)):
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" and "k_gemm" in r["Kernel_Name"]:
        cnt[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(tf)):
    if "k_gemm" in r["Kernel_Name"]:
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel                                   launches  max SQ_VALU_MFMA_BUSY_CYCLES   longest launch us   MFMA util of that launch (busy / (us x 2400 MHz x 1024 SIMDs))")
for k in sorted(cnt):
    c = max(cnt[k]); d = max(dur.get(k, [0.0]))
    print(f"{k:40s} {len(cnt[k]):8d}  {c:26.0f}   {d:17.1f}   {c / (d * 2400.0 * 1024) if d else 0:.3f}")
PY
cat $O/bench_dense.txt | grep gemm | head -8; cat $O/mfma_pmc.txt
