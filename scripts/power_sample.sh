# samples rocm-smi power / clocks while the block apply (cfg3) runs in a loop:  bash scripts/power_sample.sh [accum]
A=${1:-i8}; ST=3000; [ $A = fp64 ] && ST=700
mkdir -p gpurun_out
python bench.py --workload cfg3 --accum $A --steps $ST --warmup 2 --no-cpu-baseline --no-pca --no-alt > gpurun_out/power_bench_$A.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/^GPU\[0\]\t*: //' | tr '\n' ' '; echo
  sleep 4
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
python -c "import json; d=json.load(open('gpurun_out/power_bench_$A.json')); print('$A ms/step', d['ms_per_step'])"
