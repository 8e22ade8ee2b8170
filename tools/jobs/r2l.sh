#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
timeout 1500 python -m pytest tests/test_harness_gpu.py tests/test_distributed_gpu.py "tests/test_model_gpu.py::test_c2_full_stream_30pow4_fp64_parity" tests/test_model_gpu.py::test_float32_grid_quirk_does_not_move_the_posterior -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; grep -E "^E  |passed|failed|^>|Error|s call" $O/pytest.log | cut -c1-300 | head -40
