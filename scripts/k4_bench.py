"""K4 kernels (the eigensolver's orthogonalisation) at full height: round 4's kernels (variant 0) against the tiled ones (variant 1).
Gram = kernel + plane reduction; block GEMM = Out = Init + sum_q V_q C_q.  TB/s = basis bytes (nq blocks) / time.
usage: python scripts/k4_bench.py [N] [reps]"""
import ctypes as C
import sys

import flashpca_amd as fp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = fp.lib()
with fp.Context.synthetic(N, 512, n_pop=4, accum="fp64") as c:
    rows = c.block_rows()
    for b in (16, 32):
        for nq in (1, 2, 8, 14, 24, 27):
            line = "b=%2d nq=%2d " % (b, nq)
            for variant in (0, 1):
                L.fpca_debug_variant(0, variant)
                g, m = C.c_double(0), C.c_double(0)
                rc = L.fpca_debug_k4_bench(c.h, b, nq, reps, C.byref(g), C.byref(m))
                assert rc == 0, fp.lib().fpca_last_error()
                gb = nq * rows * b * 8 / 1e9
                line += "| v%d gram %.3f ms (%.2f TB/s)  gemm %.3f ms (%.2f TB/s) " % (variant, g.value, gb / g.value, m.value, gb / m.value)
            print(line, flush=True)
    L.fpca_debug_variant(0, 1)
    # round 6: the update of the first projection + the Gram matrices of the second in one pass (k_update_gram16), against the two launches
    for nq in (1, 2, 8, 14, 24, 27):
        g, m, f = C.c_double(0), C.c_double(0), C.c_double(0)
        assert L.fpca_debug_k4_bench(c.h, 16, nq, reps, C.byref(g), C.byref(m)) == 0
        assert L.fpca_debug_k4_fused_bench(c.h, 16, nq, reps, C.byref(f)) == 0, fp.lib().fpca_last_error()
        gb = nq * rows * 16 * 8 / 1e9
        print("b=16 nq=%2d  update %.3f + Gram %.3f = %.3f ms   fused %.3f ms (%.2f TB/s over the basis once)" % (nq, m.value, g.value, m.value + g.value, f.value, gb / f.value), flush=True)
