"""Which waves made a slow SpMV dispatch slow?  Reads the raw per-wave stamps `bench.py --stamp-dump X.npz` wrote
(wiski_prof_stamps_raw: start | placement << 48, end; 100 MHz clock) and compares the dispatches above 1.25x the median span
with the typical ones: when the waves started, how long they ran, where (XCD / CU) the last ones finished.

python tools/stamp_report.py X.npz [parts=4]"""
import sys

import numpy as np

MASK = (1 << 48) - 1


def decode(raw):
    b, e = raw[0::2], raw[1::2]
    live = e != 0
    pl = (b >> np.uint64(48)).astype(np.int64)
    t0 = (b & np.uint64(MASK)).astype(np.int64)
    t1 = e.astype(np.int64)
    return live, t0, t1, pl


def cu_key(pl):
    # simd(2) pipe(2) cu(4) sh(1) se(3) xcc(4): one id per CU = (xcc, se, sh, cu)
    return ((pl >> 12) & 15) * 1000 + ((pl >> 9) & 7) * 100 + ((pl >> 8) & 1) * 50 + ((pl >> 4) & 15)


def main():
    z = np.load(sys.argv[1])
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    D = []
    for k in sorted(z.files):
        live, t0, t1, pl = decode(z[k])
        if not live.any():
            continue
        lo = t0[live].min()
        D.append(dict(n=len(live), live=live, s=(t0 - lo) * 1e-2, e=(t1 - lo) * 1e-2, pl=pl, span=(t1[live].max() - lo) * 1e-2))
    spans = np.array([d["span"] for d in D])
    med = float(np.median(spans))
    print(f"{len(D)} dispatches, span median {med:.2f} us mean {spans.mean():.2f} us; above 1.25x median: {(spans > 1.25 * med).sum()}")
    groups = {"typical (within 3% of median)": [d for d in D if abs(d["span"] - med) <= 0.03 * med],
              "slow (> 1.25x median)": [d for d in D if d["span"] > 1.25 * med]}

    def q(v, ps=(50, 90, 99, 100)):
        return " / ".join(f"{np.percentile(v, p):6.2f}" for p in ps)

    for name, g in groups.items():
        if not g:
            continue
        print(f"\n== {name}: {len(g)} dispatches, mean span {np.mean([d['span'] for d in g]):.2f} us")
        S = np.concatenate([d["s"][d["live"]] for d in g])
        L = np.concatenate([(d["e"] - d["s"])[d["live"]] for d in g])
        E = np.concatenate([d["e"][d["live"]] for d in g])
        print(f"  wave start offset   p50/p90/p99/max  {q(S)}")
        print(f"  wave run time       p50/p90/p99/max  {q(L)}")
        print(f"  wave end offset     p50/p90/p99/max  {q(E)}")
        # per part (blockIdx.y) and per XCD (blockIdx.x % 8): the mean over dispatches of the last end / the mean run time
        for label, key in (("part", lambda d: np.arange(d["n"]) // (d["n"] // parts)), ("xcd(wg id % 8)", lambda d: (np.arange(d["n"]) % (d["n"] // parts)) % 8)):
            rows = []
            for v in range(parts if label == "part" else 8):
                last, run, st = [], [], []
                for d in g:
                    m = d["live"] & (key(d) == v)
                    last.append(d["e"][m].max()); run.append((d["e"] - d["s"])[m].mean()); st.append(d["s"][m].mean())
                rows.append(f"{v}: start {np.mean(st):5.2f} run {np.mean(run):5.2f} last {np.mean(last):5.2f}")
            print(f"  by {label}:\n    " + "\n    ".join(rows))
        # how concentrated is the lateness: per dispatch, the CUs holding the waves that finish in the last 15% of the span
        conc, ncu, wpc, late_frac = [], [], [], []
        for d in g:
            m = d["live"]
            cu = cu_key(d["pl"])[m]
            e = d["e"][m]
            late = e > 0.85 * d["span"]
            u, c = np.unique(cu, return_counts=True)
            ncu.append(len(u)); wpc.append((c.min(), np.median(c), c.max()))
            lu = np.unique(cu[late])
            conc.append(len(lu)); late_frac.append(late.mean())
        print(f"  CUs seen per dispatch: {np.mean(ncu):.1f}; waves per CU min/med/max: {np.mean([w[0] for w in wpc]):.1f} / {np.mean([w[1] for w in wpc]):.1f} / {np.mean([w[2] for w in wpc]):.1f}")
        print(f"  waves ending in the last 15% of the span: {100 * np.mean(late_frac):.1f}% of the waves, on {np.mean(conc):.1f} CUs (median {np.median(conc):.0f}, min {np.min(conc)}, max {np.max(conc)})")
    # the three slowest dispatches one by one: the 12 last waves
    for d in sorted(D, key=lambda d: -d["span"])[:3]:
        print(f"\n-- dispatch with span {d['span']:.2f} us: last 12 waves (end, start, run, part, xcc/se/sh/cu, simd)")
        idx = np.argsort(-np.where(d["live"], d["e"], -1))[:12]
        per = d["n"] // parts
        for w in idx:
            pl = int(d["pl"][w])
            print(f"   end {d['e'][w]:6.2f} start {d['s'][w]:6.2f} run {d['e'][w] - d['s'][w]:6.2f} part {w // per} wg {w % per:4d}  xcc {(pl >> 12) & 15} se {(pl >> 9) & 7} sh {(pl >> 8) & 1} cu {(pl >> 4) & 15:2d} simd {pl & 3}")
        cu = cu_key(d["pl"])
        m = d["live"]
        u = np.unique(cu[m])
        last_by_cu = np.array([d["e"][m & (cu == k)].max() for k in u])
        cnt_by_cu = np.array([(m & (cu == k)).sum() for k in u])
        o = np.argsort(-last_by_cu)
        print("   per CU last end (us), top 8: " + ", ".join(f"{u[i]}:{last_by_cu[i]:.1f}({cnt_by_cu[i]}w)" for i in o[:8]) + f";  median over CUs {np.median(last_by_cu):.1f}")
        print(f"   CUs whose last wave ends after 85% of the span: {(last_by_cu > 0.85 * d['span']).sum()} of {len(u)}")


if __name__ == "__main__":
    main()
