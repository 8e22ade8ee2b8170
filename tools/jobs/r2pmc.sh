#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --blocks 2 > /tmp/b.log 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/pm/p_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'].split('(')[0][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
tr = list(csv.DictReader(open('/tmp/pm/p_kernel_trace.csv')))
dur = collections.defaultdict(list)
for r in tr: dur[r['Kernel_Name'].split('(')[0][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, c in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    if len(dur[k]) < 10: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = sorted(dur[k])[len(dur[k]) // 2]
    print(f"{k:44s} n={len(dur[k]):4d} med {d:7.1f} us  VALU {m.get('SQ_INSTS_VALU',0):10.0f} SALU {m.get('SQ_INSTS_SALU',0):10.0f} SMEM {m.get('SQ_INSTS_SMEM',0):8.0f} waves {m.get('SQ_WAVES',0):7.0f}  SALU/CU-us {m.get('SQ_INSTS_SALU',0)/256/2400:6.1f}")
PY
