#!/bin/bash
# A/B of two builds of the int8 GEMM on the same box, interleaved (the kernel is power-limited: boxes differ by +-5 %)
for rep in 1 2 3; do
  for lib in flashpca_amd/_build/old/libfpca.so flashpca_amd/_build/libfpca.so; do
    echo -n "$lib: "; FPCA_LIB=$lib python scripts/i8_power_probe.py 2>/dev/null | head -1
  done
done
