"""Worker for tests/test_multirank_gloo.py: one rank of a world_size-N gloo job on CPU.

Each rank opens ITS SNP shard of the fileset (contiguous .bed byte range), runs the product's host
eigensolver over the host-sim backend (oracle operator on the shard) and sums the N x b partial products with a
torch.distributed(gloo) all-reduce -- the same structure bench.py / fpca_pca use on GPUs with RCCL."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.test_host_solver import hostsim  # noqa: E402


def main():
    bed, fam, k, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    replicated = len(sys.argv) > 5 and sys.argv[5] == "replicated"  # round 2's scheme: whole blocks everywhere, one all-reduce per apply
    cheap_bits = int(sys.argv[6]) if len(sys.argv) > 6 else 0       # > 0: the backend offers cheap passes (mixed-precision solver)
    tol = float(sys.argv[7]) if len(sys.argv) > 7 else 1e-8
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = O.count_fam_rows(fam)
    npk = (N + 3) // 4
    raw = np.fromfile(bed, dtype=np.uint8)[3:]
    P_total = raw.size // npk
    lo, hi = P_total * rank // world, P_total * (rank + 1) // world  # contiguous SNP range of this rank
    shard = raw[lo * npk:hi * npk].copy()
    d = O.OracleData(packed=shard, N=N, P=hi - lo, stand="binom2")

    calls = {"n": 0, "elems": 0, "small": 0, "small_elems": 0}

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64)
    def allreduce(user, buf, count):
        a = np.ctypeslib.as_array(buf, shape=(count,))
        t = torch.from_numpy(a)
        dist.all_reduce(t)  # in place on the shared memory
        if count % N == 0 and count // N in (16, 32, 48, 64):  # block-sized (N x b) sums vs the small ones (Gram coefficients, the trace)
            calls["n"] += 1
            calls["elems"] += int(count)
        else:
            calls["small"] += 1
            calls["small_elems"] += int(count)
        return 0

    L = hostsim()
    U = np.zeros((N, k), order="F")
    dv = np.zeros(k)
    Px = np.zeros((N, k), order="F")
    pve = np.zeros(k)
    tr = C.c_double()
    info = (C.c_int * 6)()
    rc = L.hostsim_pca2(d.h, k, 0, 500, tol, 2, 0, 1, 0, P_total, allreduce, None, U.ctypes.data, dv.ctypes.data,
                        Px.ctypes.data, pve.ctypes.data, C.byref(tr), info, 1 if replicated else world, rank, cheap_bits)
    # every rank must hold the same answer (replicated host algebra, deterministic)
    t = torch.from_numpy(dv.copy())
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        json.dump(dict(rc=rc, d=dv.tolist(), pve=pve.tolist(), trace=tr.value, applies=info[1], b=info[3], same=same, cheap_applies=info[4],
                       allreduce_calls=calls["n"], allreduce_elems=calls["elems"], small_calls=calls["small"],
                       small_elems=calls["small_elems"], shard=[lo, hi], P_total=P_total, N=N,
                       U0=U[:, 0].tolist(), Ulast=U[:, k - 1].tolist()), open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
