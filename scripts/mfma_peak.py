import ctypes as C, sys
sys.path.insert(0, ".")
import flashpca_amd as fp
L = fp.lib()
names = {0: "in-place, one shared A/B", 1: "in-place 4Ax2B GEMM order", 2: "in-place 4Ax2B no consecutive sharing", 3: "in-place 8 distinct A/B",
         4: "as 1, A/B at index=2 mod 4", 5: "out-of-place ping-pong (16 MFMAs/iter)"}
for pat in range(4):
    res = []
    for w in (1, 2):
        t = C.c_double()
        rc = L.fpca_debug_mfma_peak(w, 10000, pat, C.byref(t))
        v = t.value
        res.append("%d w/SIMD: %.1f TF" % (w, v))
    print("pattern %d (%s): %s" % (pat, names[pat], ", ".join(res)), flush=True)
