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


@pytest.mark.parametrize("world,k,mode", [(2, 10, "rowshard"), (3, 10, "rowshard"), (2, 20, "rowshard"),  # k = 20: two blocks of Ritz
                                          (2, 10, "replicated"), (3, 10, "replicated")])         # vectors at the automatic width 16
def test_sharded_pca_matches_golden(golden_dir, tmp_path, world, k, mode):
    name = "data_chr1"
    g = json.load(open(os.path.join(golden_dir, "golden_%s_binom2.json" % name)))
    port = free_port()
    out = str(tmp_path / "res.json")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"),
                                       os.path.join(golden_dir, name + ".bed"), os.path.join(golden_dir, name + ".fam"),
                                       str(k), out, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r = json.load(open(out))
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
