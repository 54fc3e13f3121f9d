import ctypes as C, sys, collections
sys.path.insert(0, ".")
import numpy as np
import flashpca_amd as fp
L = C.CDLL(fp.LIB_PATH)
for nwg, lds in ((256, 53248), (512, 53248), (512, 49152), (768, 49152), (1024, 20000)):
    out = np.zeros(nwg * 2, dtype=np.uint32)
    rc = L.fpca_debug_census(nwg, C.c_uint64(lds), out.ctypes.data_as(C.c_void_p))
    hw, xcc = out[0::2], out[1::2]
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; x = xcc & 0xF
    key = collections.Counter(zip(x.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    hist = collections.Counter(key.values())
    print("nwg", nwg, "lds", lds, "rc", rc, "distinct CUs", len(key), "WGs-per-CU histogram", dict(sorted(hist.items())), "per-XCD", dict(sorted(collections.Counter(x.tolist()).items())))
