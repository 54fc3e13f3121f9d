#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
