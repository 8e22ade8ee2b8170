"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, as MI355X_MICROARCH.md
prescribes) into per-kernel HBM bytes per launch.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --no-cpu-baseline --steps 5 --warmup 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --no-cpu-baseline --steps 5 --warmup 1
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a 16-B/lane coalesced stream
(the guide's correction: x2).  Only single-column (k = 1) launches of the CG kernels are kept by taking
the per-kernel median."""
import csv
import json
import statistics
import sys
from collections import defaultdict


def load(path, counter):
    d = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    f = load(fetch, "FETCH_SIZE")
    w = load(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fk = statistics.median(f[k]) if f.get(k) else 0.0
        wk = statistics.median(w[k]) if w.get(k) else 0.0
        res[k] = {"fetch_raw_kb_median": fk, "fetch_calls": len(f.get(k, [])), "write_raw_kb_median": wk, "write_calls": len(w.get(k, [])),
                  "hbm_bytes_per_launch": 2.0 * fk * 1024 + wk * 1024}
    json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) over `python bench.py --no-cpu-baseline "
                       "--steps 5 --warmup 1` on MI355X; per-launch medians in bytes. FETCH_SIZE is doubled per MI355X_MICROARCH.md "
                       "(gfx950 reports 1/2 of a 16-B/lane coalesced stream); WRITE_SIZE is uncalibrated.",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        if v["hbm_bytes_per_launch"] > 1e6:
            print(f"{k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  (fetch x2 {2 * v['fetch_raw_kb_median'] / 1e3:8.1f} MB, write {v['write_raw_kb_median'] / 1e3:8.1f} MB)")


if __name__ == "__main__":
    main()
