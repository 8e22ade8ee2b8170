"""Print the kernel timeline (start / end relative to the step's first kernel, queue) of a few streaming steps from a rocprofv3
kernel_trace.csv -- shows whether the side-stream stencil scatter of wiski_stream_step really overlaps the main stream."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34], r.get("Queue_Id", "?")) for r in rows)
g = [i for i, e in enumerate(ev) if e[2].startswith("k_gather<")]
mid = g[len(g) // 2]
i1 = g[len(g) // 2 + int(sys.argv[2]) if len(sys.argv) > 2 else len(g) // 2 + 2]
t0 = ev[mid][0]
for e in ev[mid:i1]:
    print(f"{(e[0] - t0) / 1e3:8.1f} {(e[1] - t0) / 1e3:8.1f}  {(e[1] - e[0]) / 1e3:6.1f} us  q{e[3]}  {e[2]}")
