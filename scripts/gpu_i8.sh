#!/bin/bash
# int8-sliced mode: parity tests (split-K, forced stream-K) then timing
mkdir -p gpurun_out
for k in ""; do
echo "FPCA_I8_STREAM=$k"; FPCA_I8_STREAM=$k timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "i8" 2>&1 | tail -3 | tee -a gpurun_out/i8_tests.log
done
for k in ""; do
echo "FPCA_I8_STREAM=$k"; FPCA_I8_STREAM=$k timeout 900 python scripts/i8_timing.py 2>&1 | grep "i8 {" | tee -a gpurun_out/i8_timing.log
done
