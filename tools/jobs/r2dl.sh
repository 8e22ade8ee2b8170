#!/bin/bash
for d in 12 14 16 18 20 24 28 0 16; do echo "delay $d:"; WISKI_SYM_DMA_DELAY=$d python tools/spmv_probe.py --reps 300 2>&1 | grep -E "half:"; done
