#!/bin/bash
for pm in 0120 0123 0033 1230 0233 0131; do
  echo -n "prio map $pm parts 4: "
  WISKI_HIP_SO=$PWD/build/libwiski_pm$pm.so WISKI_SYM_DMA_PARTS=4 python tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 --reps 100 2>&1 | grep "half:"
done
