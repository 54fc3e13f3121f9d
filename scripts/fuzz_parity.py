"""Randomised parity sweep (GPU): exact-integer path vs fp64 MFMA path (and, one case in four, the fp32-product mode) vs dense numpy on
random shapes, block widths, slice counts, missing-call rates -- uniform, concentrated in a few SNPs, log-normal per SNP -- forced
and automatic missing-indicator routes, and the AUTO mode's state without a sample-major copy (K2 on the int8 cores, K3 on the FP64
kernel).  python scripts/fuzz_parity.py [cases] [seed]"""
import os, sys, time
os.environ.setdefault("FPCA_LIB", "testhooks")  # the environment switches this script drives exist only in the -DFPCA_TEST_HOOKS build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LUT = {0: 2.0, 1: np.nan, 2: 1.0, 3: 0.0}


def dense_from_packed(packed, N):
    P = packed.shape[0]
    codes = np.empty((P, packed.shape[1] * 4), dtype=np.uint8)
    for s in range(4):
        codes[:, s::4] = (packed >> (2 * s)) & 3
    codes = codes[:, :N]
    G = np.select([codes == 0, codes == 2, codes == 3], [2.0, 1.0, 0.0], np.nan).T  # N x P
    n_good = np.sum(~np.isnan(G), axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.nansum(G, axis=0) / n_good
        p = mean / 2
        sd = np.sqrt(2 * p * (1 - p))
        X = (G - mean) / sd
    X[:, ~(sd > 1e-9)] = 0.0
    X[np.isnan(X)] = 0.0
    return X


worst = 0.0
t0 = time.time()
for case in range(ncases):
    big = rng.random() < 0.25
    N = int(rng.integers(1, 40000 if big else 3000))
    P = int(rng.integers(1, 30000 if big else 3000))
    if N * P > 1.2e8:
        P = max(1, int(1.2e8 // N))
    b = int(rng.choice([1, 3, 16, 20, 32, 48, 64]))
    S = int(rng.choice([7, 7, 8, 6, 5, 4]))
    rate = float(rng.choice([0.0, 0.0, 1e-4, 1e-3, 3e-3, 0.01, 0.05, 0.3]))
    # a third of the cases: missing calls CONCENTRATED in a few SNPs on top of a low uniform rate (the hybrid route: sparse
    # gathers + a compacted dense sub-matrix for the SNPs above the break-even rate)
    conc = rng.random() < 0.33
    lognorm = (not conc) and rng.random() < 0.25  # per-SNP rates exp(N(log m, sigma)): the cost model picks the route and the dense set
    npk = (N + 3) // 4
    # allele-frequency structured random codes
    maf = rng.uniform(0.0, 0.5, size=P)
    g = rng.binomial(2, maf[:, None], size=(P, npk * 4)).astype(np.uint8)
    codes = np.select([g == 2, g == 1], [0, 2], 3).astype(np.uint8)
    if conc:
        rate = float(rng.choice([0.0, 1e-4, 1e-3]))
    if lognorm:
        rate = float(rng.choice([1e-3, 5e-3, 0.02]))
        per_snp = np.minimum(0.9, rate * np.exp(rng.normal(0.0, float(rng.choice([0.5, 1.5, 2.0])), size=P)))
        codes[rng.random(codes.shape) < per_snp[:, None]] = 1
    elif rate > 0:
        codes[rng.random(codes.shape) < rate] = 1
    if conc and P >= 8:
        bad = rng.choice(P, size=max(1, int(P * rng.uniform(0.005, 0.2))), replace=False)
        for j in bad:
            codes[j, rng.random(codes.shape[1]) < rng.uniform(0.02, 0.6)] = 1
    if P > 2 and rng.random() < 0.3:
        codes[rng.integers(P)] = 1  # an all-missing SNP
        codes[rng.integers(P)] = 3  # a monomorphic SNP
    packed = (codes[:, 0::4] | (codes[:, 1::4] << 2) | (codes[:, 2::4] << 4) | (codes[:, 3::4] << 6)).astype(np.uint8)
    X = dense_from_packed(packed, N)
    B = rng.standard_normal((N, b)) * 10.0 ** rng.integers(-3, 4, size=b)
    Tin = rng.standard_normal((P, b))
    mode = int(rng.choice([-1, -1, 0, 1, 3])) if not conc else int(rng.choice([-1, -1, -1, 4, 0]))
    if lognorm:
        mode = -1
    # one case in six: the AUTO mode as it stands when the sample-major copy did not fit (test hook), default slices
    nocopy = mode < 0 and rng.random() < 0.17
    if nocopy:
        S = 7
        os.environ["FPCA_DEBUG_I8_NOCOPY"] = "1"
    else:
        os.environ.pop("FPCA_DEBUG_I8_NOCOPY", None)
    if mode >= 0:
        os.environ["FPCA_I8_MODE"] = str(mode)
    else:
        os.environ.pop("FPCA_I8_MODE", None)
    # `unit`: the expected size of the integer path's only rounding (the operand cut to 8 S - 2 bits below its column maximum) relative
    # to the result scale; PASS_FACTOR x unit is the pass threshold of one stage (the factor covers column-scale spreads of the random
    # operands and the numpy reference's own N eps), 10 x that for the composed operator.  The summary prints the worst error in units.
    unit = {8: 1e-12, 7: 1e-12, 6: 3e-11, 5: 1e-8, 4: 2e-8}[S]
    PASS_FACTOR = 50
    tol = unit
    with32 = rng.random() < 0.25
    err32 = None
    try:
        with fp.Context.from_packed(packed, N, P, accum="auto" if nocopy else "i8x%d" % S) as c8, fp.Context.from_packed(packed, N, P, accum="fp64") as c64:
            T8, T64 = c8.apply_xt(B), c64.apply_xt(B)
            Y8, Y64 = c8.apply_x(Tin), c64.apply_x(Tin)
            Z8 = c8.apply_xxt(B)
            used = c8.missing_mode(max(16, -(-b // 16) * 16))
        if with32:
            with fp.Context.from_packed(packed, N, P, accum="fp32") as c32:
                T32, Y32 = c32.apply_xt(B), c32.apply_x(Tin)
    except Exception as e:
        print("case", case, dict(N=N, P=P, b=b, S=S, rate=rate, mode=mode), "EXCEPTION", e, flush=True)
        raise
    Tr, Yr = X.T @ B, X @ Tin
    Zr = X @ Tr

    def rel(a, r):
        sc = np.max(np.abs(r), axis=0)
        sc[sc == 0] = 1.0
        return float(np.max(np.abs(a - r) / sc))

    errs = dict(T8=rel(T8, Tr), T64=rel(T64, Tr), Y8=rel(Y8, Yr), Y64=rel(Y64, Yr), Z8=rel(Z8, Zr))
    ok = errs["T8"] <= unit * PASS_FACTOR and errs["Y8"] <= unit * PASS_FACTOR and errs["Z8"] <= unit * PASS_FACTOR * 10 and errs["T64"] <= 1e-11 and errs["Y64"] <= 1e-11
    ok = ok and np.all(np.isfinite(Z8))
    if with32:  # fp32 products and short sums: ~1e-7 of sum |x||b|, measured against the largest entry of the column
        errs["T32"], errs["Y32"] = rel(T32, Tr), rel(Y32, Yr)
        ok = ok and errs["T32"] <= 3e-5 and errs["Y32"] <= 3e-5
    worst = max(worst, errs["T8"] / tol, errs["Y8"] / tol)
    if not ok or case % 10 == 0:
        print("case %3d N=%6d P=%6d b=%2d S=%d rate=%.4f%s mode=%2d(used %d)%s " % (case, N, P, b, S, rate, "+conc" if conc else "+lognormal" if lognorm else "", mode, used,
                                                                                     " no-copy" if nocopy else ""),
              " ".join("%s=%.1e" % kv for kv in errs.items()), "OK" if ok else "FAIL", flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases ok: worst stage error %.3g rounding units (pass threshold %d units), %.0f s" % (ncases, worst, 50, time.time() - t0))
