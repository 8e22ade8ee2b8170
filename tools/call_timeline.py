"""Kernel timeline of the LAST burst of GPU work in a rocprofv3 kernel_trace.csv (e.g. one model call): start, duration, gap, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]) for r in rows)
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
i = len(ev) - 1
while i > 0 and (ev[i][0] - ev[i - 1][1]) / 1e3 < gap_us:
    i -= 1
t0 = ev[i][0]
busy = 0.0
for j in range(i, len(ev)):
    s, e, n = ev[j]
    g = (s - ev[j - 1][1]) / 1e3 if j > i else 0.0
    busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  gap {g:7.1f}  {n}")
print(f"burst: {(ev[-1][1] - t0) / 1e3:.1f} us wall, {busy:.1f} us busy, {len(ev) - i} kernels")
