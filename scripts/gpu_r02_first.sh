#!/bin/bash
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_bed_upload.py tests/test_cli.py tests/test_gpu_pca.py -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r02/tests_new.log
tail -25 gpurun_out/r02/tests_new.log
nproc; free -g | head -2
timeout 900 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; tail -c 6000 gpurun_out/r02/bench_default.json; tail -5 gpurun_out/r02/bench_default.err
