"""Split the SpMV dispatches of a rocprofv3 kernel trace of bench.py by what ran before / beside them:
python tools/spmv_trace_split.py <kernel_trace.csv>"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
side = ("k_potrf_coop", "k_potrf_small", "k_tri_inv_small", "k_gram_acc", "k_basis_project", "k_gemm32", "k_woodbury_c", "k_tl_scale_cast")
side_iv = [(s, e) for s, e, n in ev if any(x in n for x in side)]
main = [(s, e, n) for s, e, n in ev if not any(x in n for x in side)]
groups = {}
for i, (s, e, n) in enumerate(main):
    if "k_spmv_sym_dma<2, true>" not in n:
        continue
    prev = main[i - 1][2].split("(")[0][:40] if i else "-"
    # position of this SpMV inside its solve: count SpMVs since the last scatter
    pos, j = 0, i - 1
    while j >= 0 and "k_scatter_stats" not in main[j][2]:
        pos += "k_spmv_sym_dma" in main[j][2]
        j -= 1
    over = any(ss < e and ee > s for ss, ee in side_iv)
    groups.setdefault((pos, over), []).append((e - s) / 1e3)
print("SpMV # in its solve | refresh kernel overlapping | n | median us | mean us | max us")
for k in sorted(groups):
    v = groups[k]
    print(f"  {k[0]:2d}  {'beside refresh' if k[1] else 'alone         '}  n={len(v):4d}  median {statistics.median(v):6.2f}  mean {statistics.mean(v):6.2f}  max {max(v):6.2f}")
allv = [x for v in groups.values() for x in v]
alone = [x for k, v in groups.items() if not k[1] for x in v]
print(f"all: n={len(allv)} mean {statistics.mean(allv):.2f} median {statistics.median(allv):.2f};  alone: n={len(alone)} mean {statistics.mean(alone):.2f} median {statistics.median(alone):.2f}")
