mkdir -p gpurun_out
# plumbing dry-run of the N=2 path on ONE GPU: gloo rendezvous, both ranks on cuda:0
FPCA_BENCH_BACKEND=gloo FPCA_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --workload tiny 2>&1 | tail -15
echo "---- single-rank RCCL communicator (nranks=1) through the library"
python - <<'PY'
import sys; sys.path.insert(0,".")
import numpy as np, flashpca_amd as fp
ctx = fp.Context.synthetic(4000, 3000, n_pop=8, accum="auto")
uid = fp.Context.comm_unique_id()
ctx.comm_init_rank(1, 0, uid)
r = ctx.pca(ndim=5)
print("nranks=1 RCCL comm ok; converged", r["info"]["converged"], "d0 %.6f" % r["d"][0])
B = np.random.default_rng(0).standard_normal((4000, 16))
ctx2 = fp.Context.synthetic(4000, 3000, n_pop=8)
print("allreduce(1 rank) identity:", np.max(np.abs(ctx.apply_xxt(B) - ctx2.apply_xxt(B))))
PY
