#!/bin/bash
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
for thr in 2e9 5e8 1e8; do
  echo "== FPCA_SPARSE_SIDE_BYTES=$thr"
  for wl in cfg4shard cfg2; do
    FPCA_SPARSE_SIDE_BYTES=$thr python bench.py --workload $wl --no-cpu-baseline --no-pca --no-alt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl', 'ms/step %.4f'%d['ms_per_step'], 'xt %.4f x %.4f gemm %.4f %.4f'%(r['ms_xt_b'],r['ms_x_t'],r['ms_gemm_kernel_xt_b'],r['ms_gemm_kernel_x_t']))"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace_cfg4shard_i8 -o bench -- python bench.py --workload cfg4shard --no-cpu-baseline --no-alt --no-pca > /dev/null 2>&1
f=$(ls gpurun_out/r02/trace_cfg4shard_i8/*kernel_stats.csv | head -1); cut -c1-100,200-400 $f | head -14
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02/trace_cfg4shard_i8/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70].ljust(70), r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
