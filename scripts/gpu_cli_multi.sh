# the --gpus launcher on ONE GPU through the host-shared-memory test transport: outputs must equal the single-process run
set +e
B=$GRAFT_REPO_ROOT/flashpca_amd/_build/flashpca; D=$GRAFT_REPO_ROOT/tests/golden/hapmap3_data
rm -rf /tmp/m1 /tmp/m2 /tmp/m3; mkdir -p /tmp/m1 /tmp/m2 /tmp/m3
(cd /tmp/m1 && $B --bfile $D --ndim 10 --outload load.txt --outmeansd ms.txt --precision 12 -v > log.txt 2>&1; tail -3 log.txt)
(cd /tmp/m2 && FPCA_CLI_TEST_TRANSPORT=shm $B --bfile $D --ndim 10 --gpus 2 --outload load.txt --outmeansd ms.txt --precision 12 -v > log.txt 2>&1; echo "rc=$?"; grep -E "GPUs|applies|Exception" log.txt)
(cd /tmp/m3 && FPCA_CLI_TEST_TRANSPORT=shm $B --bfile $D --ndim 10 --gpus 3 --outload load.txt --outmeansd ms.txt --precision 12 > log.txt 2>&1; echo "rc=$?"; grep -E "GPUs|applies|Exception" log.txt || true)
python - <<'PY'
import numpy as np
def tab(p, skip=2):
    return np.array([l.split("\t")[skip:] for l in open(p).read().splitlines()[1:]], dtype=float)
for d in ("/tmp/m2", "/tmp/m3"):
    e1, e2 = np.loadtxt("/tmp/m1/eigenvalues.txt"), np.loadtxt(d + "/eigenvalues.txt")
    U1, U2 = tab("/tmp/m1/eigenvectors.txt"), tab(d + "/eigenvectors.txt")
    V1, V2 = tab("/tmp/m1/load.txt"), tab(d + "/load.txt")
    m1, m2 = tab("/tmp/m1/ms.txt"), tab(d + "/ms.txt")
    sg = np.sign(np.sum(U1 * U2, axis=0))
    print(d, "eigenvalues %.1e  eigenvectors %.1e  loadings %.1e  meansd %.1e  pve %.1e" % (
        np.max(np.abs(e1 - e2) / e1), np.max(np.abs(U1 - U2 * sg)), np.max(np.abs(V1 - V2 * sg)), np.nanmax(np.abs(m1 - m2)),
        np.max(np.abs(np.loadtxt("/tmp/m1/pve.txt") - np.loadtxt(d + "/pve.txt")))))
PY
# the real transport on one GPU must fail cleanly (RCCL refuses duplicate devices... with --gpus 2 the second rank asks for device 1)
(cd /tmp/m2 && $B --bfile $D --ndim 10 --gpus 2 > log2.txt 2>&1; echo "no second GPU: rc=$?"; tail -3 log2.txt)
