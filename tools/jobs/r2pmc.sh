#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 --k 64 --reps 3 > /tmp/b.log 2>&1
  python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/pm/p_counter_collection.csv')))
agg = collections.defaultdict(list)
for r in rows:
    if 'k_spmm_sym_cols' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items(): print(f"{k:32s} per launch {sum(v)/len(v):16.0f}  (n={len(v)})")
tr = list(csv.DictReader(open('/tmp/pm/p_kernel_trace.csv')))
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tr if 'k_spmm_sym_cols' in r['Kernel_Name']]
print("kernel duration us:", [x / 1e3 for x in d])
PY
done
