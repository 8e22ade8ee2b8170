#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
