"""Is the int8 GEMM power-bound or latency-bound?  Same kernel, same memory traffic, same instruction stream -- but an
all-zero fp64 operand B (all slices zero: the multipliers toggle nothing).  A latency-bound kernel does not care; a
power-bound one speeds up like the bare MFMA stream does (3470 -> 4540 TOP/s)."""
import sys
sys.path.insert(0, ".")
import torch
import flashpca_amd as fp
N, P, b = 500000, 100000, 32
with fp.Context.synthetic(N, P, n_pop=40, accum="i8", missing_rate=0.0) as ctx:
    ctx.stats()
    rows = ctx.block_rows()
    Y = torch.zeros((rows, b), dtype=torch.float64, device="cuda")
    for what in ("random", "zero", "tiny (one nonzero entry per column)", "random"):
        B = torch.zeros((rows, b), dtype=torch.float64, device="cuda")
        if what == "random":
            B[:N] = torch.rand((N, b), dtype=torch.float64, device="cuda") - 0.5
        elif what.startswith("tiny"):
            B[0, :] = 1.0
        torch.cuda.synchronize()
        for _ in range(3):
            ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
        ctx.synchronize()
        ctx.profile_begin(10, sample_every=1)
        for _ in range(10):
            ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
        p = ctx.profile_end(b)
        print("B = %-40s K2 GEMM %.3f ms   K3 GEMM %.3f ms" % (what, p["ms_gemm_xt"], p["ms_gemm_x"]), flush=True)
