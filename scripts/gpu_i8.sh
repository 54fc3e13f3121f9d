#!/bin/bash
# int8-sliced mode: parity tests then timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "i8" 2>&1 | tail -15 | tee gpurun_out/i8_tests.log
timeout 900 python scripts/i8_timing.py 2>&1 | tee gpurun_out/i8_timing.log
