#!/bin/bash
# int8-sliced mode: parity tests then timing
mkdir -p gpurun_out
for sh in D C; do
FPCA_I8_SHAPE=$sh timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "i8" 2>&1 | tail -3 | tee -a gpurun_out/i8_tests.log
done
for sh in C; do
echo "shape $sh"; FPCA_I8_SHAPE=$sh timeout 900 python scripts/i8_timing.py cfg3 2>&1 | tee -a gpurun_out/i8_timing.log
done
timeout 900 python scripts/i8_timing.py cfg2 2>&1 | tee -a gpurun_out/i8_timing.log
