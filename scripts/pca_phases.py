"""Where the wall-clock of one fpca_pca call goes (FPCA_TIMING=1 prints the phases on stderr): python scripts/pca_phases.py [N P k]"""
import os, sys, time
os.environ["FPCA_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

N, P, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (500000, 100000, 20)
with fp.Context.synthetic(N, P, n_pop=min(2 * k, 64), accum="auto") as ctx:
    ctx.stats()
    for i in range(3):
        t = time.perf_counter()
        r = ctx.pca(ndim=k)
        ctx.synchronize()
        w = time.perf_counter() - t
        info = r["info"]
        print("call %d: wall %.4f s  apply %.4f  ortho %.4f  host %.4f  total(C) %.4f  applies %d" % (
            i, w, info["seconds_apply"], info["seconds_ortho"], info["seconds_host"], info["seconds_total"], info["block_applies"]), file=sys.stderr)

# the Python side of one call, piece by piece (what is NOT inside fpca_pca)
import ctypes as C
from flashpca_amd._lib import PcaOpts, PcaInfo, lib
with fp.Context.synthetic(N, P, n_pop=min(2 * k, 64), accum="auto") as ctx:
    ctx.stats()
    ctx.pca(ndim=k)
    keep = None
    for i in range(4):
        T = [time.perf_counter()]
        o = PcaOpts(); lib().fpca_pca_init_opts(C.byref(o), C.sizeof(PcaOpts), C.sizeof(PcaInfo)); o.ndim = k
        U = np.empty((N, k), order="F"); d = np.empty(k); Px = np.empty((N, k), order="F"); pve = np.empty(k); ms = np.empty((P, 2), order="F")
        info = PcaInfo()
        T.append(time.perf_counter())
        rc = lib().fpca_pca(ctx.h, C.byref(o), U.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), Px.ctypes.data_as(C.c_void_p),
                            pve.ctypes.data_as(C.c_void_p), None, ms.ctypes.data_as(C.c_void_p), C.byref(info))
        T.append(time.perf_counter())
        ctx.synchronize()
        T.append(time.perf_counter())
        keep = (U, Px)  # drops the previous call's arrays here
        T.append(time.perf_counter())
        print("py %d: alloc %.3f ms | fpca_pca %.3f ms (C total %.3f) | synchronize %.3f | free previous %.3f" % (
            i, (T[1] - T[0]) * 1e3, (T[2] - T[1]) * 1e3, info.seconds_total * 1e3, (T[3] - T[2]) * 1e3, (T[4] - T[3]) * 1e3), file=sys.stderr)
