#!/bin/bash
# A/B of the sparse gather kernels on one box: stage times at cfg3 / cfg2 / cfg4shard, gathers inline (exact cost) and as shipped
for side in 1e30 default; do
  for g in 1 2; do
    for wl in cfg3 cfg2 cfg4shard; do
      if [ $side = default ]; then unset FPCA_SPARSE_SIDE_BYTES; else export FPCA_SPARSE_SIDE_BYTES=$side; fi
      FPCA_LIB=testhooks FPCA_GATHER=$g python bench.py --workload $wl --no-cpu-baseline --no-alt --no-pca 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('side=$side gather=$g $wl ms/step %.4f  K2 stage %.4f (gemm %.4f)  K3 stage %.4f (gemm %.4f)'%(d['ms_per_step'], r['ms_xt_b'], r['ms_gemm_kernel_xt_b'], r['ms_x_t'], r['ms_gemm_kernel_x_t']))"
    done
  done
done
