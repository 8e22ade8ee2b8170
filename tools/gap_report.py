"""GPU-busy fraction and the largest inter-kernel gaps of a streaming run, from a rocprofv3 kernel_trace.csv
(`rocprofv3 --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-extras`)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]) for r in rows)
sc = [i for i, e in enumerate(ev) if e[2].startswith("k_scatter_stats_sym")]
a, b = sc[len(sc) // 4], sc[-len(sc) // 4]
busy = sum(e[1] - e[0] for e in ev[a:b])
span = ev[b][0] - ev[a][0]
# gaps above 1 ms are pass boundaries of bench.py (model rebuilt from the init data, un-timed): reported, but kept out of the busy fraction
big = sum(c[0] - p[1] for p, c in zip(ev[a:b], ev[a + 1:b + 1]) if c[0] - p[1] > 1_000_000)
nst = max(1, len([i for i in sc if a <= i < b]))
print(f"steps {nst}: {(span - big) / 1e3 / nst:.1f} us per step under the tracer (pass-boundary gaps of {big / 1e6:.1f} ms excluded), GPU busy {busy / (span - big):.3f}")
gaps = {}
for p, c in zip(ev[a:b], ev[a + 1:b + 1]):
    gaps.setdefault((p[2][:22], c[2][:22]), []).append((c[0] - p[1]) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  {k[0]:24s} -> {k[1]:24s} n {len(v):5d}  mean gap {sum(v) / len(v):6.1f} us  total {sum(v) / 1e3:7.2f} ms")
