#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_spectral_gpu.py tests/test_two_level_gpu.py tests/test_hip_ops.py -q -x 2>&1 | tail -2
timeout 300 python tools/tl_cold_probe.py 2>&1 | grep -v amdgpu > $O/tl_cold.txt; grep "64 columns\|one column\|rebuild" $O/tl_cold.txt
cd /tmp; rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o v -- python $R/tools/tl_cold_probe.py > /dev/null 2>&1; cp /tmp/pv/v_kernel_stats.csv $O/variance_kernel_stats.csv; cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/variance_kernel_stats.csv')))
for r in rows[:9]: print(f"{float(r['AverageNs'])/1e3:8.1f} us x {int(r['Calls']):4d}  {r['Name'][:80]}")
PY
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; r=json.load(open('$O/bench.json')); e=r['extra']
print(r['value'], r['ms_per_step'], e['cg_iters_per_step_mean'], {k: round(e[k],3) for k in e if k.startswith('variance_ms') or k.startswith('reference_step_ms')})"
