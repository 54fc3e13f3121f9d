#!/bin/bash
mkdir -p gpurun_out/r02
python scripts/cpu_threads_probe.py 2>&1 | tee gpurun_out/r02/cpu_threads_probe.txt
for wl in cfg4shard cfg5shard; do
  python bench.py --workload $wl --no-cpu-baseline --no-pca-hard > gpurun_out/r02/bench_${wl}_n1.json 2> gpurun_out/r02/bench_${wl}_n1.err; tail -c 2500 gpurun_out/r02/bench_${wl}_n1.json; tail -2 gpurun_out/r02/bench_${wl}_n1.err
done
python bench.py --workload cfg5shard --accum fp32 --no-cpu-baseline --no-pca-hard --no-alt > gpurun_out/r02/bench_cfg5shard_n1_fp32.json 2>/dev/null; tail -c 1500 gpurun_out/r02/bench_cfg5shard_n1_fp32.json
echo "---- torchrun 2 ranks on a 1-GPU box: must fail cleanly, not hang"
( time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --workload tiny ) 2>&1 | tail -25 | tee gpurun_out/r02/torchrun_2ranks_on_1gpu.txt
echo "---- remaining GPU test files"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dense.py tests/test_reference_hapmap3_script.py tests/test_reference_testthat_check_project.py tests/test_reference_testthat_pca.py tests/test_cabi.py -x -q -m gpu 2>&1 | tail -5
