"""Kernel timeline of ONE reference-fidelity step from a rocprofv3 kernel_trace.csv of tools/refstep_trace.py: start (us), gap to the
previous kernel, duration, name -- steps are delimited by k_eig_update launches; the sixth step from the end is printed."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]) for r in rows)
# steps delimited by k_basis_eig_update
marks = [i for i, e in enumerate(ev) if e[2].startswith("k_eig_update")]
a, b = marks[-6], marks[-5]
t0 = ev[a][0]
print("step span us", (ev[b][0] - t0) / 1e3, "kernels", b - a, "busy us", sum(e[1] - e[0] for e in ev[a:b]) / 1e3)
prev_end = t0
for s, e, n in ev[a:b]:
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:8.1f}  +{gap:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n}")
    prev_end = e
