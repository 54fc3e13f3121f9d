"""Round 6: sustained rate of bare FP matrix-instruction streams (no memory traffic) with zero and with pseudo-random operands, and the
package power / shader clock while they run -- the practical ceilings the fp32 / fp64 GEMM kernels' roofline fractions are read against.
usage: python scripts/mfma_fp_peak.py"""
import ctypes as C
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import flashpca_amd as fp

L = fp.lib()
names = {20: "f32 16x16x4 zeros", 21: "f32 16x16x4 random", 22: "f32 32x32x2 zeros", 23: "f32 32x32x2 random", 24: "f64 16x16x4 zeros", 25: "f64 16x16x4 random",
         10: "i8 32x32x32 zeros", 11: "i8 32x32x32 random"}
peak = {20: 157.3, 21: 157.3, 22: 157.3, 23: 157.3, 24: 78.6, 25: 78.6, 10: 5033.0, 11: 5033.0}


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        w = [l.split(":")[-1].strip() for l in r.splitlines() if "Package Power" in l]
        c = [l for l in r.splitlines() if "sclk" in l]
        if w and c:
            out.append((float(w[0]), c[0].split("(")[1].split("Mhz")[0]))


for pat in (20, 21, 22, 23, 24, 25, 10, 11):
    for w in (1, 2):
        t = C.c_double()
        iters = 400000 if pat >= 20 else 300000
        stop, smp = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, smp))
        th.start()
        t0 = time.time()
        best = 0.0
        while time.time() - t0 < 2.5:
            fp._lib.check(L.fpca_debug_mfma_peak(w, iters, pat, C.byref(t)))
            best = max(best, t.value)
        stop.set()
        th.join()
        busy = [s for s in smp if s[0] > 600]
        print("%-20s %d wave/SIMD: %8.1f = %.3f of %.1f   power %s W  sclk %s MHz" % (
            names[pat], w, best, best / peak[pat], peak[pat], (sorted(b[0] for b in busy)[len(busy) // 2] if busy else "-"),
            (sorted(int(b[1]) for b in busy)[len(busy) // 2] if busy else "-")), flush=True)
# the fp32 / fp64 block applies at 16 columns: power and clock while they loop
for accum in ("fp32", "fp64"):
    with fp.Context.synthetic(500000, 100000, n_pop=40, accum=accum) as c:
        c.bench_apply(b=16, steps=3, warmup=2)
        stop, smp = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, smp))
        th.start()
        r = c.bench_apply(b=16, steps=80 if accum == "fp32" else 40, warmup=2)
        stop.set()
        th.join()
        busy = [s for s in smp if s[0] > 600]
        fl = 2.0 * 500000 * 100000 * 16
        pk = 157.3 if accum == "fp32" else 78.6
        print("%s apply b=16: K2 %.3f ms (%.3f)  K3 %.3f ms (%.3f)   power %s W  sclk %s MHz (%d samples)" % (
            accum, r["ms_gemm_xt"], fl / r["ms_gemm_xt"] / 1e9 / pk, r["ms_gemm_x"], fl / r["ms_gemm_x"] / 1e9 / pk,
            (sorted(b[0] for b in busy)[len(busy) // 2] if busy else "-"), (sorted(int(b[1]) for b in busy)[len(busy) // 2] if busy else "-"), len(busy)), flush=True)
