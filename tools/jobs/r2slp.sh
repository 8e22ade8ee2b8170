#!/bin/bash
for so in online_gp_amd/csrc/libwiski_hip.so build/libwiski_noslp.so; do
  echo "== $so"
  export WISKI_HIP_SO=$PWD/$so
  python bench.py --no-cpu-baseline --no-extras --blocks 40 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
"
  python tools/var_probe.py 64 2>&1 | grep variance | tail -1
  python tools/hyper_probe.py 20 2>&1 | grep _hyper_step
done
