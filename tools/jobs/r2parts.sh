#!/bin/bash
for p in 4 5 6 7; do
  echo "== parts $p"
  WISKI_SYM_DMA_PARTS=$p python tools/spmv_probe.py --dim 3 --grid 50 --dtype f32 2>&1 | grep -v amdgpu | tail -3
done
