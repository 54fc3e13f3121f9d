"""world_size-2 (and 3) CPU test of the SNP-sharded path over torch.distributed/gloo: SNP shards + the sum of the N x b
partial products give the single-rank answer (svdwide.cpp:48-62 sums SNP blocks the same way) -- in the row-sharded solver
(the default with several ranks: all-gather -> operator -> reduce-scatter, every rank orthogonalises N / G rows of the basis,
the Gram coefficients are the only other collective) and in round 2's replicated one (one all-reduce per apply)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, bed, fam, k, out, mode, extra=()):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), bed, fam, str(k), out, mode] + [str(x) for x in extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return json.load(open(out))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_small_sample_path(tmp_path, world):
    """Fewer than three blocks of samples (N = 40, k = 8): the solver forms X X' from applies on the identity (dense_small)
    -- upload, apply and download are all collectives in the row-sharded backend, and every rank must walk through them
    alike."""
    from oracle import oracle as O

    N, P, k = 40, 301, 8
    rng = np.random.default_rng(world)
    packed = rng.integers(0, 256, size=(P, (N + 3) // 4), dtype=np.uint8)
    bed, fam = str(tmp_path / "t.bed"), str(tmp_path / "t.fam")
    with open(bed, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(packed.tobytes())
    open(fam, "w").write("".join("F%d I%d 0 0 0 -9\n" % (i, i) for i in range(N)))
    r = run_world(world, bed, fam, k, str(tmp_path / "res.json"), "rowshard")
    assert r["rc"] == 0 and r["same"]
    X = O.OracleData(packed=packed, N=N, P=P, stand="binom2").dense()
    w = np.linalg.eigvalsh(X @ X.T / P)[::-1][:k]
    assert np.max(np.abs(np.array(r["d"]) - w)) < 1e-10 * w[0]
    assert r["applies"] == -(-N // r["b"])


@pytest.mark.parametrize("world,k,mode", [(2, 10, "rowshard"), (3, 10, "rowshard"), (2, 20, "rowshard"),  # k = 20: two blocks of Ritz
                                          (2, 10, "replicated"), (3, 10, "replicated")])         # vectors at the automatic width 16
def test_sharded_pca_matches_golden(golden_dir, tmp_path, world, k, mode):
    name = "data_chr1"
    g = json.load(open(os.path.join(golden_dir, "golden_%s_binom2.json" % name)))
    out = str(tmp_path / "res.json")
    r = run_world(world, os.path.join(golden_dir, name + ".bed"), os.path.join(golden_dir, name + ".fam"), k, out, mode)
    assert r["rc"] == 0 and r["same"]
    ev = np.array(g["eigenvalues_div_p"])[:k]
    assert np.max(np.abs(np.array(r["d"]) - ev) / ev) < 1e-9  # divisor uses the TOTAL SNP count
    assert abs(r["trace"] * r["P_total"] - g["trace_raw"]) <= 1e-12 * g["trace_raw"]  # scalar all-reduce of the trace
    assert np.max(np.abs(np.array(r["pve"]) - np.array(g["pve"])[:k])) < 1e-11
    U0 = np.array(g["U_first5"])[0]
    assert abs(abs(U0 @ np.array(r["U0"])) - 1) < 1e-8
    assert abs(np.linalg.norm(r["Ulast"]) - 1) < 1e-10  # (row-sharded: every rank's U went through the final all-gather)
    nb = g["N"] * r["b"]
    if mode == "replicated":
        # exactly one all-reduce of N x b per block apply (+ one scalar for the trace): no other data-path collective
        assert (r["allreduce_calls"], r["allreduce_elems"]) == (r["applies"], r["applies"] * nb)
        assert (r["small_calls"], r["small_elems"]) == (1, 1)
    else:
        # per block apply: one all-gather + one reduce-scatter of the N x b block (each built here from the transport's
        # sum; over RCCL the two move exactly the bytes of the one all-reduce they replace) + ceil(k/b) all-gathers of the
        # Ritz blocks for the download; everything else is small: three Gram sums of (m+1) b^2 coefficients per step, two
        # for the start block, the trace
        kb = -(-k // r["b"])
        assert r["allreduce_calls"] == 2 * r["applies"] + kb and r["allreduce_elems"] == (2 * r["applies"] + kb) * nb
        assert r["small_calls"] >= 3 * r["applies"] + 3
        assert r["small_elems"] <= (3 * r["applies"] + 8) * (r["applies"] + 1) * r["b"] ** 2 + 1


@pytest.mark.parametrize("world,mode", [(2, "rowshard"), (3, "rowshard"), (2, "replicated")])
def test_sharded_mixed_precision_solver(golden_dir, tmp_path, world, mode):
    """The mixed-precision solver with several ranks (k = 30 at tol 1e-6 on data_chr1: slow enough for cheap passes): every decision
    -- exact -> cheap passes, the verification through the exact operator, its verdict -- is taken from Ritz data that is
    identical on all ranks, so the ranks walk through the same sequence of collectives; the answer is the dense one."""
    from oracle import oracle as O

    name, k = "data_chr1", 30
    bed, fam = os.path.join(golden_dir, name + ".bed"), os.path.join(golden_dir, name + ".fam")
    r = run_world(world, bed, fam, k, str(tmp_path / "res.json"), mode, extra=(30, 1e-6))
    assert r["rc"] == 0 and r["same"]
    assert 0 < r["cheap_applies"] < r["applies"]
    N = O.count_fam_rows(fam)
    d = O.OracleData(bed, N, "binom2")
    X = d.dense()
    w = np.linalg.eigvalsh(X @ X.T)[::-1][:k] / d.P
    assert np.max(np.abs(np.array(r["d"]) - w) / w) < 1e-9

