#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/i8_full_tests.log
for a in i8 i8x6; do
timeout 600 python bench.py --accum $a --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg2_$a.json; cat gpurun_out/bench_cfg2_$a.json | cut -c1-1500
timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 1 --accum $a --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_$a.json; cat gpurun_out/bench_cfg3_$a.json | cut -c1-1500
done
timeout 1200 python bench.py --workload cfg5 --steps 3 --warmup 1 --accum i8 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cfg5_i8.json; cat gpurun_out/bench_cfg5_i8.json | cut -c1-1500
