# flashpca --gpus G with all G ranks on ONE GPU over the host-memory test transport (the RCCL-shaped call sequence: all-gather ->
# K2, K3 chunk by chunk -> reduce-scatter), against the one-process run on the same fileset:
#   bash scripts/gpu_cli_multi.sh [N] [P] [G] [rowshard|replicated]      (a plumbing check at size, not a measurement)
mkdir -p /tmp/e2m && cd /tmp/e2m
N=${1:-500000}; P=${2:-100000}; G=${3:-8}; S=${4:-rowshard}
python $GRAFT_REPO_ROOT/scripts/make_synth_bed.py $N $P /tmp/e2m/syn | tail -1
mkdir -p one multi
( cd one && time $GRAFT_REPO_ROOT/flashpca_amd/_build/flashpca --bfile /tmp/e2m/syn --ndim 20 --precision 12 -v ) 2>&1 | grep -i "eigensolver\|block applies\|real\|error" | tail -6
( cd multi && time FPCA_CLI_TEST_TRANSPORT=shm2 $GRAFT_REPO_ROOT/flashpca_amd/_build/testhooks/flashpca --bfile /tmp/e2m/syn --ndim 20 --precision 12 --gpus $G --solver $S -v ) 2>&1 | grep -i "eigensolver\|block applies\|real\|error\|fpca\]" | tail -12
python - <<PY
import numpy as np
a, b = np.loadtxt("one/eigenvalues.txt"), np.loadtxt("multi/eigenvalues.txt")
print("eigenvalues, max rel diff %d ranks vs 1: %.2e" % ($G, np.max(np.abs(a - b) / a)))
def tab(p):
    return np.array([[float(x) for x in l.split()[2:]] for l in open(p).read().splitlines()[1:]])
U, V = tab("one/eigenvectors.txt"), tab("multi/eigenvectors.txt")
sg = np.sign(np.sum(U * V, axis=0))
print("eigenvectors, max abs diff: %.2e" % np.max(np.abs(U - V * sg)))
PY
rm -rf /tmp/e2m
