set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
ls -la gpurun_out/prof_r1
head -30 gpurun_out/prof_r1/*kernel_stats.csv
python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_cfg3.err | tee gpurun_out/bench_cfg3.json
tail -3 gpurun_out/bench_cfg3.err
