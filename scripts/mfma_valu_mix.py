"""Round 6: what do VALU / LDS instructions cost a matrix-instruction stream?  v_mfma_f32_16x16x4_f32 (32 cycles) with VPM independent
v_add_u32 per MFMA, interleaved or in bursts of 8 MFMAs (the GEMM kernels' shape), 1 / 2 / 4 waves per SIMD.  Fraction of 157.3 TFLOP/s."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import flashpca_amd as fp

L = fp.lib()
for ldsr in (0, 1):
    for burst in (0, 1):
        for vpm in ((0, 1, 2, 3, 4, 6, 8) if ldsr == 0 else (0, 2, 3)):
            if vpm == 0 and burst:
                continue
            row = []
            for w in (1, 2, 4):
                t = C.c_double()
                fp._lib.check(L.fpca_debug_mfma_peak(w, 200000, 1000 + 100 * ldsr + 10 * vpm + burst, C.byref(t)))
                row.append("%d w/SIMD %.3f" % (w, t.value / 157.3))
            print("LDS reads/MFMA %d  VALU/MFMA %d  %-11s: %s" % (ldsr, vpm, "bursts of 8" if burst else "interleaved", "   ".join(row)), flush=True)
