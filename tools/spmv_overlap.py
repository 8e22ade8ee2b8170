"""Which kernels ran beside the slow SpMV dispatches?  rocprofv3 kernel trace of bench.py -> for every k_spmv_sym_dma<2, true>
dispatch the kernels (any stream) whose [start, end] intersects it; per overlapping kernel name: how many dispatches it met, how
many of those were slow (> 1.25x the median), the mean duration of the dispatches it met.
python tools/spmv_overlap.py <kernel_trace.csv>"""
import csv
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in rows), key=lambda e: e[0])
sp = [(s, e) for s, e, n in ev if "k_spmv_sym_dma<2, true>" in n]
dur = [(e - s) / 1e3 for s, e in sp]
med = statistics.median(dur)
slow = [d > 1.25 * med for d in dur]
print(f"{len(sp)} SpMV dispatches, median {med:.2f} us, mean {statistics.mean(dur):.2f} us, slow: {sum(slow)}")
met = defaultdict(list)
lonely_slow = 0
for (s, e), d, sl in zip(sp, dur, slow):
    names = {n for ss, ee, n in ev if ss < e and ee > s and not (ss == s and ee == e)}
    if sl and not names:
        lonely_slow += 1
    for n in names:
        met[n].append((d, sl))
    if not names:
        met["(nothing)"].append((d, sl))
print(f"{'overlapping kernel':46s} met  slow  mean us of the SpMVs it met")
for n, v in sorted(met.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    print(f"{n:46s} {len(v):4d} {sum(x[1] for x in v):4d}  {statistics.mean(x[0] for x in v):6.2f}")
