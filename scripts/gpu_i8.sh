#!/bin/bash
# int8-sliced mode: parity tests then timing
mkdir -p gpurun_out
for sh in D; do
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "i8" 2>&1 | tail -3 | tee -a gpurun_out/i8_tests.log
done
timeout 900 python -m pytest tests/test_gpu_pca.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/i8_timing.py 2>&1 | tee -a gpurun_out/i8_timing.log
