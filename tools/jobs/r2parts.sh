#!/bin/bash
for dp in 2 3 4; do for i in 1 2; do echo -n "depth $dp: "; WISKI_HIP_SO=$PWD/build/libwiski_reg$dp.so WISKI_SYM_REG=1 python tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 --reps 200 2>&1 | grep -E "half:" ; done; WISKI_HIP_SO=$PWD/build/libwiski_reg$dp.so WISKI_SYM_REG=1 python tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 --reps 5 2>&1 | grep -E "diff"; done
