"""The b = 16 column remainder on the matrix pipe (fpca_debug_mfma_peak patterns 12 / 13): useful TOP/s of a wave's MFMAs for
S b = 112 slice-columns as four 32-wide tiles (half of the last one zero padding) vs three tiles + v_mfma_i32_16x16x64_i8."""
import ctypes as C, sys
sys.path.insert(0, ".")
import flashpca_amd as fp
L = fp.lib()
for rep in range(3):
    res = []
    for pat, nm in ((12, "4 x 32-wide tiles"), (13, "3 tiles + 16x16x64"), (11, "bare 32x32x32 stream")):
        t = C.c_double()
        fp._lib.check(L.fpca_debug_mfma_peak(1, 100000 if pat >= 12 else 300000, pat, C.byref(t)))
        res.append("%s: %.0f" % (nm, t.value))
    print("useful TOP/s  " + " | ".join(res), flush=True)
