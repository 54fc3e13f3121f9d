#!/bin/bash
# int8 GEMMs: parity, timing and FETCH_SIZE (L2 -> fabric read traffic) at cfg3
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "i8" 2>&1 | tail -2
timeout 900 python scripts/i8_timing.py 2>&1 | grep "i8 {\|i8x6 {"
rm -rf gpurun_out/fetch_i8
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_i8 -o pmc -- python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-pca --no-alt > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fetch_i8/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_gemm_i8" in n:
        agg["K3" if "I8Cfg<true" in n else "K2"].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, "FETCH_SIZE x2 = %.2f GB per launch" % (sum(v) / len(v) * 1024 * 2 / 1e9))
PY
