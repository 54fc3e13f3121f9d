#!/bin/bash
mkdir -p gpurun_out
python scripts/mfma_i8_peak.py 2>&1 | tee gpurun_out/i8_peak.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/run_i8.py <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import flashpca_amd as fp
ctx = fp.Context.synthetic(500000, 100000, n_pop=40, accum="i8")
r = ctx.bench_apply(b=32, steps=3, warmup=1)
print(r)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/i8prof -- python /tmp/run_i8.py 2>&1 | tail -3
cd /root/repo
f=$(ls gpurun_out/i8prof/*/*kernel_stats.csv | head -1); cat $f | head -20 | tee gpurun_out/i8_kernel_stats.csv
