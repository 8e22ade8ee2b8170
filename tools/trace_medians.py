"""Median / min kernel durations from a rocprofv3 kernel_trace.csv (medians are robust against the few
multi-column launches of the bench's un-timed extras)."""
import csv
import statistics
import sys
from collections import defaultdict

d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5:
        print(f"{k[:70]:70s} n={len(v):4d} median {statistics.median(v):8.1f} us  min {min(v):8.1f}  total {sum(v) / 1e3:8.2f} ms")
