import ctypes as C, sys
sys.path.insert(0, ".")
import flashpca_amd as fp
L = fp.lib()
for pat, nm in ((10, "zero operands"), (11, "random operands")):
    res = []
    for w in (1, 2, 4):
        t = C.c_double()
        rc = L.fpca_debug_mfma_peak(w, 20000, pat, C.byref(t))
        res.append("%d w/SIMD: %.0f TOP/s" % (w, t.value))
    print("v_mfma_i32_32x32x32_i8 (%s): %s" % (nm, ", ".join(res)), flush=True)
