"""GPU parity tests of the individual kernels, through the C ABI, against the CPU oracle (oracle/) and the
committed goldens (tests/golden/).  Tolerances: integer / table work bit-exact; fp64 GEMM outputs 1e-12
relative to the column norm (fp64 accumulation order differs between an MFMA tree and the oracle's loops).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-12


@pytest.fixture(scope="module")
def fp(built_lib):
    import flashpca_amd

    return flashpca_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as O

    O.build()
    return O


def _load(golden_dir, name, stand, fp, orc):
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    bed = os.path.join(golden_dir, name + ".bed")
    ctx = fp.Context.from_bed(bed, N, stand=stand)
    od = orc.OracleData(bed, N, stand)
    g = json.load(open(os.path.join(golden_dir, "golden_%s_%s.json" % (name, stand))))
    return ctx, od, g


def test_device_is_gfx950(fp):
    import ctypes as C

    L = fp.lib()
    assert L.fpca_device_count() >= 1
    buf = C.create_string_buffer(256)
    assert L.fpca_device_name(0, buf, 256) == 0
    assert b"gfx950" in buf.value


def test_mfma_operand_mapping(fp):
    """A=asymmetric, B=asymmetric: catches a transposed or permuted C/D mapping (guide section 3)."""
    import ctypes as C

    rng = np.random.default_rng(7)
    A = rng.standard_normal((16, 4))
    B = rng.standard_normal((4, 16))
    D = np.zeros((16, 16))
    rc = fp.lib().fpca_debug_mfma_probe(A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert np.allclose(D, A @ B, rtol=0, atol=1e-14)


@pytest.mark.parametrize("name", ["data_chr1", "hapmap3_data"])
@pytest.mark.parametrize("stand", ["binom2", "binom"])
def test_stats_bit_exact(golden_dir, name, stand, fp, orc):
    """K1: mean and sd must equal the oracle bit for bit (integer counts, one division, one sqrt)."""
    ctx, od, g = _load(golden_dir, name, stand, fp, orc)
    ms, trace = ctx.stats()
    od.dense()  # visits every SNP -> fills the oracle's mean/sd
    oms = od.meansd()
    assert ctx.N == g["N"] and ctx.P == g["P"]
    assert np.array_equal(ms[:, 0], oms[:, 0])
    assert np.array_equal(ms[:, 1], oms[:, 1])
    assert np.allclose(ms[:8, 0], g["mean_first8"], rtol=0, atol=0)
    assert abs(trace - g["trace_raw"]) <= 1e-12 * g["trace_raw"]
    ctx.close()


@pytest.mark.parametrize("name,b", [("data_chr1", 1), ("data_chr1", 5), ("data_chr1", 16), ("data_chr1", 37),
                                    ("hapmap3_data", 32), ("hapmap3_data", 64), ("hapmap3_data", 70)])
def test_operator_parity(golden_dir, name, b, fp, orc):
    """K2, K3 and their composition vs the oracle's dense block products, every block width class."""
    ctx, od, g = _load(golden_dir, name, "binom2", fp, orc)
    rng = np.random.default_rng(b)
    X = od.dense()  # N x P standardised, from the oracle (data.cpp:215-335 restated)
    B = rng.standard_normal((ctx.N, b))
    T_ref = X.T @ B
    T = ctx.apply_xt(B)
    assert np.max(np.abs(T - T_ref)) <= REL * np.max(np.abs(T_ref)) * 10
    Tin = rng.standard_normal((ctx.P, b))
    Y_ref = X @ Tin
    Y = ctx.apply_x(Tin)
    assert np.max(np.abs(Y - Y_ref)) <= REL * np.max(np.abs(Y_ref)) * 10
    Z_ref = X @ T_ref
    Z = ctx.apply_xxt(B)
    assert np.max(np.abs(Z - Z_ref)) <= REL * np.max(np.abs(Z_ref)) * 10
    ctx.close()


def test_operator_probe_golden(golden_dir, fp, orc):
    """y = X X' probe against the numpy golden (independent of the C oracle)."""
    for name in ("data_chr1", "hapmap3_data"):
        ctx, od, g = _load(golden_dir, name, "binom2", fp, orc)
        probe = np.cos(0.37 * np.arange(ctx.N) + 0.11) + 0.25
        y = ctx.apply_xxt(probe.reshape(-1, 1))[:, 0]
        assert np.allclose(y[:8], g["probe_y_first8"], rtol=1e-11, atol=0)
        assert abs(np.linalg.norm(y) - g["probe_y_norm"]) <= 1e-12 * g["probe_y_norm"]
        ctx.close()


@pytest.mark.parametrize("N,P", [(1, 3), (5, 7), (64, 1), (257, 300), (1000, 513), (2051, 129)])
def test_ragged_shapes_vs_oracle(N, P, fp, orc):
    """Edge shapes: N not a multiple of 4 (pad bits in the last byte), tiny / single SNP, monomorphic and
    all-missing SNPs."""
    rng = np.random.default_rng(N * 1000 + P)
    npk = (N + 3) // 4
    packed = rng.integers(0, 256, size=(P, npk), dtype=np.uint8)
    if P >= 3:
        packed[1, :] = 0xFF  # monomorphic (all hom A2) -> zero column (data.cpp:299)
        packed[2, :] = 0x55  # all missing -> NaN mean, zero column
    ctx = fp.Context.from_packed(packed, N, P)
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    X = od.dense()
    ms, _ = ctx.stats()
    oms = od.meansd()
    assert np.array_equal(np.isnan(ms), np.isnan(oms))
    assert np.array_equal(ms[~np.isnan(ms)], oms[~np.isnan(oms)])
    B = rng.standard_normal((N, 3))
    Z_ref = X @ (X.T @ B)
    Z = ctx.apply_xxt(B)
    assert np.all(np.isfinite(Z))
    assert np.max(np.abs(Z - Z_ref)) <= 1e-11 * max(1.0, np.max(np.abs(Z_ref)))
    ctx.close()


def test_synthetic_generator_and_roundtrip(fp, orc):
    """The on-GPU generator is deterministic, shard-consistent, and its matrix round-trips through the oracle."""
    N, P = 3000, 700
    a = fp.Context.synthetic(N, P, snp_begin=0, n_pop=12)
    b1 = fp.Context.synthetic(N, 300, snp_begin=0, n_pop=12)
    b2 = fp.Context.synthetic(N, 400, snp_begin=300, n_pop=12)
    pa = a.download_packed().reshape(P, -1)
    assert np.array_equal(pa[:300], b1.download_packed().reshape(300, -1))
    assert np.array_equal(pa[300:], b2.download_packed().reshape(400, -1))
    codes = np.stack([(pa >> (2 * s)) & 3 for s in range(4)], axis=-1).reshape(P, -1)[:, :N]
    miss = (codes == 1).mean()
    assert 0.0002 < miss < 0.003  # missing_rate 0.001
    od = orc.OracleData(packed=pa, N=N, P=P, stand="binom2")
    X = od.dense()
    rng = np.random.default_rng(0)
    B = rng.standard_normal((N, 32))
    Z = a.apply_xxt(B)
    Z_ref = X @ (X.T @ B)
    assert np.max(np.abs(Z - Z_ref)) <= 1e-11 * np.max(np.abs(Z_ref))
    # sharded sum == full (the multi-GPU identity, svdwide.cpp:48-62)
    Zs = b1.apply_xxt(B) + b2.apply_xxt(B)
    assert np.max(np.abs(Zs - Z)) <= 1e-11 * np.max(np.abs(Z))
    for c in (a, b1, b2):
        c.close()


def test_linearity_at_scale(fp):
    """Size-independent property at a size the oracle cannot touch: X X'(a u + v) == a X X' u + X X' v."""
    N, P = 50000, 20000
    ctx = fp.Context.synthetic(N, P)
    rng = np.random.default_rng(1)
    u = rng.standard_normal((N, 16))
    v = rng.standard_normal((N, 16))
    lhs = ctx.apply_xxt(2.5 * u + v)
    rhs = 2.5 * ctx.apply_xxt(u) + ctx.apply_xxt(v)
    assert np.max(np.abs(lhs - rhs)) <= 1e-11 * np.max(np.abs(lhs))
    # symmetry: u' (A v) == (A u)' v
    Au, Av = ctx.apply_xxt(u), ctx.apply_xxt(v)
    s1, s2 = np.sum(u * Av), np.sum(Au * v)
    assert abs(s1 - s2) <= 1e-10 * abs(s1)
    ctx.close()


@pytest.mark.parametrize("name,b", [("data_chr1", 16), ("hapmap3_data", 32), ("hapmap3_data", 48), ("hapmap3_data", 64)])
def test_fp32_mode_operator_tolerance(golden_dir, name, b, fp, orc):
    """FPCA_ACCUM_FP32 (BASELINE config 5 "fp32 accumulate"): fp32 MFMA products, fp32 sums within four chunks (512 samples / 256 SNPs), fp64 across
    chunks.  Tolerance: 2e-6 of the output scale per entry (fp32 rounding of the table and of B is 6e-8 relative per
    product; the observed error is ~1e-7)."""
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    bed = os.path.join(golden_dir, name + ".bed")
    ctx = fp.Context.from_bed(bed, N, accum="fp32")
    od = orc.OracleData(bed, N, "binom2")
    X = od.dense()
    rng = np.random.default_rng(b)
    B = rng.standard_normal((N, b))
    T_ref = X.T @ B
    T = ctx.apply_xt(B)
    e_t = np.max(np.abs(T - T_ref)) / np.max(np.abs(T_ref))
    Z_ref = X @ T_ref
    Z = ctx.apply_xxt(B)
    e_z = np.max(np.abs(Z - Z_ref)) / np.max(np.abs(Z_ref))
    assert 1e-12 < e_t < 2e-6 and 1e-12 < e_z < 2e-6, (e_t, e_z)  # really fp32, and within tolerance
    ms, _ = ctx.stats()
    od.dense()
    assert np.array_equal(ms, od.meansd())  # statistics stay fp64 / bit-exact in both modes
    ctx.close()


def test_mfma_i8_operand_mapping(fp):
    """v_mfma_i32_32x32x32_i8 lane->operand map of kernels_i8.hip (asymmetric operands, full int8 range)."""
    import ctypes as C

    rng = np.random.default_rng(11)
    A = rng.integers(-128, 128, size=(32, 32), dtype=np.int8)
    Bt = rng.integers(-128, 128, size=(32, 32), dtype=np.int8)
    D = np.zeros((32, 32), dtype=np.int32)
    rc = fp.lib().fpca_debug_mfma_i8_probe(A.ctypes.data_as(C.c_void_p), Bt.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert np.array_equal(D, A.astype(np.int32) @ Bt.astype(np.int32).T)


@pytest.mark.parametrize("name,b,S,tol", [("data_chr1", 32, 7, 1e-12), ("hapmap3_data", 32, 7, 1e-12), ("hapmap3_data", 64, 7, 1e-12),
                                          ("data_chr1", 32, 8, 1e-12), ("hapmap3_data", 64, 8, 1e-12), ("hapmap3_data", 16, 7, 1e-12),
                                          ("hapmap3_data", 16, 8, 1e-12), ("hapmap3_data", 48, 7, 1e-11), ("data_chr1", 5, 8, 1e-12),
                                          ("hapmap3_data", 64, 4, 3e-6), ("hapmap3_data", 32, 6, 1e-9)])
def test_i8_mode_operator_parity(golden_dir, name, b, S, tol, fp, orc):
    """FPCA_ACCUM_I8(S): integer genotype matrices x byte slices of the fp64 operand with exact int32 accumulation.
    S = 7 keeps 54 bits per column scale, so the result meets the fp64 tolerance; smaller S only truncates the operand
    (error <= 2^-(8S-1) of the column maximum per element, no accumulation error)."""
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    bed = os.path.join(golden_dir, name + ".bed")
    ctx = fp.Context.from_bed(bed, N, accum="i8x%d" % S)
    od = orc.OracleData(bed, N, "binom2")
    X = od.dense()
    rng = np.random.default_rng(b + S)
    B = rng.standard_normal((N, b)) * np.logspace(-3, 3, b)[None, :]  # per-column scales must not matter
    T_ref = X.T @ B
    T = ctx.apply_xt(B)
    assert np.max(np.abs(T - T_ref) / np.max(np.abs(T_ref), axis=0)) <= 10 * tol
    Tin = rng.standard_normal((ctx.P, b))
    Y_ref = X @ Tin
    Y = ctx.apply_x(Tin)
    assert np.max(np.abs(Y - Y_ref) / np.max(np.abs(Y_ref), axis=0)) <= 10 * tol
    Z_ref = X @ T_ref
    Z = ctx.apply_xxt(B)
    assert np.max(np.abs(Z - Z_ref) / np.max(np.abs(Z_ref), axis=0)) <= 10 * tol
    ctx.close()


@pytest.mark.parametrize("S", [7, 4])
def test_i8_slicing_extreme_columns(golden_dir, S, fp, orc):
    """The per-column fixed-point slicing at its corners: columns scaled to the ends of the double range, a zero column,
    a single-entry column, a constant column (every element = the column maximum, top byte at its bound), all-negative
    and alternating +-max columns (carries into the top byte), and denormals.  Each column of the result must carry its
    own relative accuracy, untouched by its neighbours."""
    name = "data_chr1"
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    bed = os.path.join(golden_dir, name + ".bed")
    od = orc.OracleData(bed, N, "binom2")
    X = od.dense()
    rng = np.random.default_rng(S)
    b = 16
    B = rng.standard_normal((N, b))
    B[:, 0] *= 1e-290
    B[:, 1] *= 1e290
    B[:, 2] = 0.0
    B[:, 3] = 0.0
    B[N // 2, 3] = -3.75
    B[:, 4] = 0.999999999999
    B[:, 5] = -np.abs(B[:, 5]) - 1.0
    B[:, 6] = np.where(np.arange(N) % 2 == 0, 1.0, -1.0) * (2.0 - 2.0 ** -52)
    B[:, 7] *= 5e-324 * 2 ** 20  # denormals
    B[:, 8] = 2.0 ** -1030  # a constant denormal column
    with fp.Context.from_bed(bed, N, accum="i8x%d" % S) as ctx:
        T = ctx.apply_xt(B)
    T_ref = X.T @ B
    # the integer path is exact up to the 2^-(8S-1) rounding of each operand entry relative to its column maximum; the
    # numpy reference itself carries ~N eps of the absolute sums -- so the bound is on |X|' |B| (several columns cancel
    # almost completely: a constant column against centred genotypes), column by column
    with np.errstate(under="ignore"):
        A = np.abs(X).T @ np.maximum(np.abs(B), np.max(np.abs(B), axis=0, keepdims=True) * 2.0 ** -(8 * S - 2))
    for c in range(b):
        err = np.abs(T[:, c] - T_ref[:, c])
        if not np.any(B[:, c]):
            assert np.all(T[:, c] == 0.0), c
        elif c in (7, 8):
            assert np.all(err <= 1e-9 * A[:, c] + 5e-324 * N * 4096), c  # the denormal grid is the limit
        else:
            assert np.all(err <= (2.0 ** -(8 * S - 3) + 1e-13) * A[:, c]), (c, np.max(err / A[:, c]))
    assert np.all(np.isfinite(T))


@pytest.mark.parametrize("S", [7, 5, 4])  # (4: the sparse gathers read fp32 rows scaled by the column's slice exponent)
@pytest.mark.parametrize("preloaded", [False, True])
def test_i8_slicing_extreme_columns_k3_side(S, preloaded, fp, orc):
    """The K3 twin of the test above: Y = X T where the int8 operands are T / sd and mean T / sd (kernels_i8.hip k_slice with a
    per-ROW scale) -- T columns at the ends of the double range / zero / single-entry / constant / alternating, AND rows whose
    scale 1/sd spans orders of magnitude: the rare-variant spectrum of the realistic generator (sd 0.03 .. 0.71, singletons and
    SNPs monomorphic in this sample -> zero rows, data.cpp:299-320), and, preloaded (data.cpp:293-297, the --project route:
    dense missing-indicator kernels), sd from 1e-4 to 1 with means anywhere in [0, 2].  Against the oracle's dense matrix,
    column by column, each with its own relative accuracy."""
    N, P, b = 3001, 2000, 16
    rng = np.random.default_rng(100 + S)
    with fp.Context.synthetic(N, P, n_pop=5, realistic=True, accum="fp64") as gen:
        packed = gen.download_packed().reshape(P, -1).copy()
    packed[7, :] = 0xFF   # a SNP that is monomorphic in this sample (every dosage 0): sd = 0 -> a zero column (data.cpp:299-320)
    packed[11, :] = 0x55  # ... and one with every call missing (mean = NaN)
    packed = packed.reshape(-1)
    with fp.Context.from_packed(packed, N, P, accum="i8x%d" % S) as ctx:
        od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
        if preloaded:
            ms = np.column_stack([rng.uniform(0.0, 2.0, P), 10.0 ** rng.uniform(-4, 0, P)])
            ms[:5, 1] = [1e-4, 1.0, 3e-4, 0.5, 1e-3]
            od.set_preloaded_meansd(ms)
            ctx.set_meansd(ms)
        X = od.dense()
        if not preloaded:
            sd = od.meansd()[:, 1]
            assert np.sum(~(sd > 1e-9)) >= 2 and np.nanmin(sd[sd > 1e-9]) < 0.05 and np.nanmax(sd) > 0.69  # the spread this test is about
        T = rng.standard_normal((P, b))
        T[:, 0] *= 1e-290
        T[:, 1] *= 1e280
        T[:, 2] = 0.0
        T[:, 3] = 0.0
        T[P // 3, 3] = -3.75
        T[:, 4] = 0.999999999999
        T[:, 5] = -np.abs(T[:, 5]) - 1.0
        T[:, 6] = np.where(np.arange(P) % 2 == 0, 1.0, -1.0) * (2.0 - 2.0 ** -52)
        T[:, 7] *= 2.0 ** -1000
        Y = ctx.apply_x(T)
        assert ctx.missing_mode(b) == (0 if preloaded else 4)  # preloaded statistics: two-matrix kernels; else the hybrid route
    Y_ref = X @ T
    # what the integer path rounds is T / sd and mean T / sd, each entry by at most 2^-(8S-1) of ITS operand's column maximum;
    # pushed through the integer matrices (dosage <= 2, indicator <= 1, P terms per row) that bounds the error of a row by
    # 2^-(8S-1) P (2 max|T/sd| + max|mean T/sd|); the numpy reference adds its own ~P eps of |X| |T|
    msd = od.meansd()
    isd = np.where(msd[:, 1] > 1e-9, 1.0 / np.where(msd[:, 1] > 1e-9, msd[:, 1], 1.0), 0.0)
    with np.errstate(under="ignore", over="ignore"):
        for c in range(b):
            tg, tm = np.abs(T[:, c]) * isd, np.abs(T[:, c]) * isd * np.abs(np.nan_to_num(msd[:, 0]))
            bound = 2.0 ** -(8 * S - 2) * P * (2.0 * tg.max() + tm.max()) + 1e-13 * (np.abs(X) @ np.abs(T[:, c]))
            err = np.abs(Y[:, c] - Y_ref[:, c])
            if not np.any(T[:, c]):
                assert np.all(Y[:, c] == 0.0), c
            else:
                assert np.all(err <= bound + 5e-324), (c, float(np.max(err / np.maximum(bound, 1e-300))))
                # ... and the bound is not vacuous: a column's result is resolved to well below its own scale
                assert np.max(bound) < 1e-6 * np.max(np.abs(Y_ref[:, c])) * (256.0 ** (7 - S)), c
    assert np.all(np.isfinite(Y))


@pytest.mark.parametrize("S", [7, 4])
def test_i8_hybrid_missing_route(S, fp, orc):
    """Missing calls concentrated in few SNPs (the realistic generator: 5 % of the SNPs lose 10-30 % of their calls, the rest
    <= 0.1 %; ~1 % overall, above the sparse route's break-even): per SNP choice of the route -- the dense SNPs'
    indicator matrix as a compacted integer GEMM, everybody else's missing calls as sparse gathers, data.cpp:300-320 ([1] -> 0)
    either way.  All three products at 16 / 32 / 64 columns against the oracle's dense matrix; then a width without a gather
    kernel (48) on the same context, which must fall back to the two-matrix kernels WITH the plain sample-major copy, and the
    widths again after that."""
    N, P = 5003, 3001
    with fp.Context.synthetic(N, P, n_pop=5, realistic=True, accum="i8x%d" % S) as ctx:
        packed = ctx.download_packed()
        od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
        X = od.dense()
        nmiss = np.sum(X == 0.0, axis=0)  # (includes monomorphic columns; only the spread matters here)
        assert np.sum(nmiss > 0.05 * N) > 0.02 * P and np.median(nmiss) < 0.005 * N
        rng = np.random.default_rng(S)
        tol = 1e-11 if S == 7 else 1e-7  # (S = 4: 30-bit operands, rounded once per operand and stage)
        fell_back = False
        for b in (16, 32, 64, 48, 16):
            B = rng.standard_normal((N, b))
            Tin = rng.standard_normal((P, b))
            fell_back = fell_back or b == 48
            mode = ctx.missing_mode(b)
            assert mode == (0 if fell_back else 4), (b, mode)
            Z, T, Y = ctx.apply_xxt(B), ctx.apply_xt(B), ctx.apply_x(Tin)
            Zr, Tr, Yr = X @ (X.T @ B), X.T @ B, X @ Tin
            assert np.max(np.abs(Z - Zr)) <= tol * np.max(np.abs(Zr)), (b, mode)
            assert np.max(np.abs(T - Tr)) <= tol * np.max(np.abs(Tr)), (b, mode)
            assert np.max(np.abs(Y - Yr)) <= tol * np.max(np.abs(Yr)), (b, mode)
        assert ctx.missing_mode(16) == 0  # after the fallback the context stays on the dense route
        ms, tr = ctx.stats()
        assert np.array_equal(ms, od.meansd(), equal_nan=True)
        assert np.array_equal(ctx.download_packed(), packed)  # the records whose missing calls were masked are back, bit for bit


@pytest.mark.parametrize("mean,sigma,expect", [(0.02, 1.5, 4), (0.01, 2.0, 4), (0.02, 0.3, 0), (0.001, 0.5, 3)])
def test_missing_route_follows_the_cost_model_on_lognormal_rates(mean, sigma, expect, fp, orc):
    """Round 5: the route of the missing-call indicator is chosen from a cost model over K1's per-SNP counts (missing_routes.hip
    hybrid_classify), and the generator has a profile between "uniform" and "5 % of the SNPs hold nearly everything": per-SNP rates
    log-normally distributed.  A long tail (sigma 1.5-2) sends the shard down the hybrid route with a LARGE dense set -- 30-50 % of
    the SNPs, far beyond round 4's "at most a quarter" -- a narrow spread around 2 % stays on the two-matrix kernels like a uniform
    rate, a low rate on the sparse route; every product against the oracle's dense matrix (data.cpp:300-320: [1] -> 0) either way."""
    N, P = 20000, 4096
    with fp.Context.synthetic(N, P, n_pop=6, missing_rate=mean, missing_model=2, lognormal_sigma=sigma, accum="i8") as ctx:
        packed = ctx.download_packed()
        od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
        X = od.dense()
        codes = np.stack([(packed.reshape(P, -1) >> (2 * s)) & 3 for s in range(4)], axis=-1).reshape(P, -1)[:, :N]
        rate = (codes == 1).mean(axis=1)
        assert abs(rate.mean() - mean) < 0.25 * mean and (sigma < 1 or rate.max() > 8 * np.median(rate))  # the profile is what it says
        assert ctx.missing_mode(16) == expect, (ctx.missing_mode(16), rate.mean(), float(np.mean(rate > 0.0069)))
        if expect == 4:
            assert np.mean(rate > 0.0069) > (0.3 if mean >= 0.02 else 0.15)  # (round 4 admitted a dense set of at most a quarter, and only with the rest below 0.5 %)
        rng = np.random.default_rng(int(1000 * sigma))
        for b in (16, 32):
            B = rng.standard_normal((N, b))
            Tin = rng.standard_normal((P, b))
            Z, T, Y = ctx.apply_xxt(B), ctx.apply_xt(B), ctx.apply_x(Tin)
            Zr, Tr, Yr = X @ (X.T @ B), X.T @ B, X @ Tin
            assert np.max(np.abs(Z - Zr)) <= 1e-11 * np.max(np.abs(Zr)), b
            assert np.max(np.abs(T - Tr)) <= 1e-11 * np.max(np.abs(Tr)), b
            assert np.max(np.abs(Y - Yr)) <= 1e-11 * np.max(np.abs(Yr)), b
        assert np.array_equal(ctx.download_packed(), packed)
        r = ctx.pca(ndim=5)
        w = np.linalg.eigvalsh(X.T @ X if P < N else X @ X.T)[::-1][:5] / P
        assert np.max(np.abs(r["d"] - w) / w) < 1e-8


@pytest.mark.parametrize("kw,expect_mode", [(dict(missing_rate=0.001), 3), (dict(missing_rate=0.02), 0), (dict(realistic=True), 0), (dict(missing_rate=0.0), 2)])
def test_auto_keeps_k2_on_the_int8_cores_when_only_the_second_copy_does_not_fit(kw, expect_mode, fp, orc, monkeypatch):
    """FPCA_ACCUM_AUTO on the largest inputs (round 5): when the sample-major copy of the packed matrix is what does not fit, only
    K3 -- the stage that reads it -- drops to the FP64-MFMA kernel; K2 reads the SNP-major matrix the context holds anyway and stays on
    the int8 matrix cores (rounds 1-4 dropped BOTH stages: 4.6 x the time where the matrix is biggest).  Forced here by making that
    one allocation fail (test build); all three products and a solve against the oracle, every missing-call route that state allows
    (the hybrid route lives on a view of the copy: those data take the two-matrix K2)."""
    N, P = 5003, 3001
    with fp.Context.synthetic(N, P, n_pop=5, accum="fp64", **kw) as ref:
        packed = ref.download_packed()
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    X = od.dense()
    monkeypatch.setenv("FPCA_DEBUG_I8_NOCOPY", "1")
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=5, accum="auto", **kw) as ctx:
        rng = np.random.default_rng(11)
        for b in (16, 32, 64, 20):
            B = rng.standard_normal((N, b))
            Tin = rng.standard_normal((P, b))
            Z, T, Y = ctx.apply_xxt(B), ctx.apply_xt(B), ctx.apply_x(Tin)
            Zr, Tr, Yr = X @ (X.T @ B), X.T @ B, X @ Tin
            assert np.max(np.abs(Z - Zr)) <= 1e-11 * np.max(np.abs(Zr)), b
            assert np.max(np.abs(T - Tr)) <= 1e-11 * np.max(np.abs(Tr)), b
            assert np.max(np.abs(Y - Yr)) <= 1e-11 * np.max(np.abs(Yr)), b
        assert ctx.accum == "i8x7" and ctx.missing_mode(16) == expect_mode  # still the exact-integer context: K2's route
        r = ctx.pca(ndim=6)
        w = np.linalg.eigvalsh(X.T @ X)[::-1][:6] / P
        assert np.max(np.abs(r["d"] - w) / w) < 1e-8
        assert np.array_equal(ctx.download_packed(), packed)


@pytest.mark.parametrize("N,P", [(1, 3), (5, 7), (257, 300), (2051, 129)])
def test_i8_mode_ragged_shapes(N, P, fp, orc):
    rng = np.random.default_rng(N * 1000 + P)
    npk = (N + 3) // 4
    packed = rng.integers(0, 256, size=(P, npk), dtype=np.uint8)
    if P >= 3:
        packed[1, :] = 0xFF
        packed[2, :] = 0x55
    ctx = fp.Context.from_packed(packed, N, P, accum="i8")
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    X = od.dense()
    B = rng.standard_normal((N, 3))
    Z_ref = X @ (X.T @ B)
    Z = ctx.apply_xxt(B)
    assert np.all(np.isfinite(Z))
    assert np.max(np.abs(Z - Z_ref)) <= 1e-11 * max(1.0, np.max(np.abs(Z_ref)))
    ctx.close()


def test_auto_mode_resolution(golden_dir, fp):
    """FPCA_ACCUM_AUTO: the exact-integer path for 2-bit input; explicit modes are reported as given."""
    N = fp.count_fam_rows(os.path.join(golden_dir, "data_chr1.fam"))
    bed = os.path.join(golden_dir, "data_chr1.bed")
    with fp.Context.from_bed(bed, N, accum="auto") as c:
        assert c.accum == "i8x7"
    with fp.Context.from_bed(bed, N, accum="fp64") as c:
        assert c.accum == "fp64"
    with fp.Context.from_bed(bed, N, accum="i8x6") as c:
        assert c.accum == "i8x6"
    with fp.Context.from_dense(np.random.default_rng(0).standard_normal((50, 20)), stand="sd") as c:
        assert c.accum == "fp64"


@pytest.mark.parametrize("accum", ["fp64", "auto"])
def test_allreduce_hook_and_native_single_rank(fp, accum):
    """The two multi-GPU transports on one rank: the caller-supplied hook (called once per block apply with the N_pad x b
    product, and it wins over a built-in communicator) and the built-in RCCL communicator with nranks = 1."""
    N, P = 3000, 900
    rng = np.random.default_rng(5)
    B = rng.standard_normal((N, 16))
    with fp.Context.synthetic(N, P, n_pop=6, accum=accum) as ref:
        Z0 = ref.apply_xxt(B)
    calls = []
    with fp.Context.synthetic(N, P, n_pop=6, accum=accum) as c:
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        Z1 = c.apply_xxt(B)  # the int8 path computes and all-reduces Y in row chunks here (other split-K plan per chunk)
        assert np.max(np.abs(Z1 - Z0)) <= 1e-13 * np.max(np.abs(Z0))

        def hook(ptr, count, stream):
            calls.append(count)
            return 0

        c.set_allreduce(hook)
        assert np.array_equal(c.apply_xxt(B), Z0)
        assert len(calls) == 1 and calls[0] == c.block_rows() * 16
        assert c.P_total == P


@pytest.mark.parametrize("accum", ["fp64", "auto"])
def test_empty_shard(fp, accum):
    """A rank may own zero SNPs (more GPUs than SNP tiles): its partial product is exactly zero."""
    N = 300
    with fp.Context.from_packed(np.zeros((0, (N + 3) // 4), dtype=np.uint8), N, 0, accum=accum) as c:
        B = np.random.default_rng(0).standard_normal((N, 4))
        Z = c.apply_xxt(B)
        assert Z.shape == (N, 4) and np.all(Z == 0.0)
        ms, trace = c.stats()
        assert ms.shape[0] == 0 and trace == 0.0


def test_auto_mode_falls_back_to_fp64_when_buffers_do_not_fit(fp, monkeypatch):
    """FPCA_ACCUM_AUTO switches to the fp64 kernels (and says so) if the int8 path cannot allocate its buffers; an
    explicitly requested int8 mode reports the error instead."""
    N, P = 2000, 700
    B = np.random.default_rng(1).standard_normal((N, 8))
    with fp.Context.synthetic(N, P, n_pop=6, accum="fp64") as ref:
        Z0 = ref.apply_xxt(B)
    monkeypatch.setenv("FPCA_DEBUG_I8_NOMEM", "1")
    with fp.Context.synthetic(N, P, n_pop=6, accum="auto") as c:  # the shipped library has no such switch
        Z = c.apply_xxt(B)
        assert c.accum == "i8x7"
    with fp.test_hooks():  # the -DFPCA_TEST_HOOKS build of the same sources
        with fp.Context.synthetic(N, P, n_pop=6, accum="auto") as c:
            assert c.accum == "i8x7"
            Z = c.apply_xxt(B)
            assert c.accum == "fp64"
            assert np.array_equal(Z, Z0)
        with fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:
            with pytest.raises(fp.FpcaError) as ei:
                c.apply_xxt(B)
            assert ei.value.code == -4 and "--accum auto" in str(ei.value) and "--gpus" in str(ei.value)  # says what would fit


def test_sparse_missing_route_falls_back_when_its_lists_do_not_fit(golden_dir, fp, orc, monkeypatch):
    """The sparse missing-indicator route needs 8 bytes per missing call; if those lists do not fit the context switches to
    the dense route (both integer matrices on the matrix cores) and says so, instead of failing the apply."""
    N = fp.count_fam_rows(os.path.join(golden_dir, "hapmap3_data.fam"))
    bed = os.path.join(golden_dir, "hapmap3_data.bed")
    X = orc.OracleData(bed, N, "binom2").dense()
    B = np.random.default_rng(9).standard_normal((N, 16))
    Z_ref = X @ (X.T @ B)
    monkeypatch.setenv("FPCA_DEBUG_SPARSE_NOMEM", "1")
    with fp.test_hooks(), fp.Context.from_bed(bed, N, accum="i8") as ctx:
        assert ctx.missing_mode(16) == 3  # 0.15 % missing calls: the sparse route is the automatic choice
        Z = ctx.apply_xxt(B)
        assert ctx.missing_mode(16) in (0, 1) and ctx.accum == "i8x7"  # still the exact-integer path, dense indicator
        assert np.max(np.abs(Z - Z_ref) / np.max(np.abs(Z_ref), axis=0)) <= 1e-11
        Y = ctx.apply_x(X.T @ B)
        assert np.max(np.abs(Y - Z_ref) / np.max(np.abs(Z_ref), axis=0)) <= 1e-11
    # the same for the hybrid route (whose sample-major copy is a VIEW without the dense SNPs' missing calls: the fallback must
    # put the plain copy back before the two-matrix kernels read it)
    N2, P2 = 3001, 2000
    with fp.Context.synthetic(N2, P2, n_pop=5, realistic=True, accum="fp64") as ref:
        B2 = np.random.default_rng(10).standard_normal((N2, 16))
        Z2 = ref.apply_xxt(B2)
    with fp.test_hooks(), fp.Context.synthetic(N2, P2, n_pop=5, realistic=True, accum="i8") as ctx:
        assert ctx.missing_mode(16) == 4
        Z = ctx.apply_xxt(B2)
        assert ctx.missing_mode(16) in (0, 1)
        assert np.max(np.abs(Z - Z2)) <= 1e-11 * np.max(np.abs(Z2))


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
def test_overlapped_allreduce_row_chunks(fp, monkeypatch, nch):
    """Built-in communicator: K3 + all-reduce in 1..4 row chunks (communication stream, events) give the same Y."""
    N, P = 40000, 1500
    B = np.random.default_rng(2).standard_normal((N, 32))
    with fp.Context.synthetic(N, P, n_pop=6, accum="i8") as ref:
        Z0 = ref.apply_xxt(B)
    monkeypatch.setenv("FPCA_AR_CHUNKS", str(nch))
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        assert c.allreduce_chunks() == nch  # the setting reached the library (it is read per call, not cached)
        for _ in range(3):
            Z = c.apply_xxt(B)
            assert np.max(np.abs(Z - Z0)) <= 1e-13 * np.max(np.abs(Z0))
        r = c.pca(ndim=5)
        assert r["info"]["converged"] == 1


@pytest.mark.parametrize("N,P,b", [(3001, 1999, 32), (2050, 700, 64), (517, 300, 16)])
def test_i8_shard_without_missing_genotypes(N, P, b, fp, orc):
    """No missing call in the shard (imputed / 1000-Genomes-like data): E = 0, only G.M is multiplied and M'Q = 1'Q comes
    from the column sums; the padding rows / samples (all "missing") must still come out exactly zero."""
    rng = np.random.default_rng(N)
    with fp.Context.synthetic(N, P, n_pop=6, missing_rate=0.0, accum="i8") as c:
        packed = c.download_packed().reshape(P, -1)
        od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
        X = od.dense()
        codes = np.stack([(packed >> (2 * s)) & 3 for s in range(4)], axis=-1).reshape(P, -1)[:, :N]
        assert not (codes == 1).any()
        B = rng.standard_normal((N, b))
        T_ref = X.T @ B
        assert np.max(np.abs(c.apply_xt(B) - T_ref) / np.max(np.abs(T_ref), axis=0)) <= 1e-11
        Tin = rng.standard_normal((P, b))
        Y_ref = X @ Tin
        assert np.max(np.abs(c.apply_x(Tin) - Y_ref) / np.max(np.abs(Y_ref), axis=0)) <= 1e-11
        Z_ref = X @ T_ref
        assert np.max(np.abs(c.apply_xxt(B) - Z_ref) / np.max(np.abs(Z_ref), axis=0)) <= 1e-11
        r = c.pca(ndim=5)
        w = np.linalg.eigvalsh(X @ X.T)[::-1][:5] / P
        assert np.max(np.abs(r["d"] - w) / w) < 1e-8


@pytest.mark.parametrize("mode", ["0", "1", "3"])
def test_i8_forced_missing_modes(golden_dir, fp, orc, monkeypatch, mode):
    """Both matrices on the matrix cores (0), the same skipping blocks of the missing indicator without a missing
    genotype (1), and G.M on the matrix cores + the missing-indicator products as sparse fp64 gathers (3) give the same
    operator on data WITH missing calls, whatever the automatic choice would be."""
    monkeypatch.setenv("FPCA_I8_MODE", mode)
    N = fp.count_fam_rows(os.path.join(golden_dir, "hapmap3_data.fam"))
    bed = os.path.join(golden_dir, "hapmap3_data.bed")
    od = orc.OracleData(bed, N, "binom2")
    X = od.dense()
    with fp.Context.from_bed(bed, N, accum="i8") as shipped:  # the product ignores the variable: its own choice (sparse)
        assert shipped.missing_mode(32) == 3
    with fp.test_hooks(), fp.Context.from_bed(bed, N, accum="i8") as ctx:
        for b in (32, 64, 16, 5):
            assert ctx.missing_mode(b) == int(mode)  # HapMap3 has 0.15 % missing calls: every forced path applies
            B = np.random.default_rng(3).standard_normal((N, b))
            T_ref = X.T @ B
            assert np.max(np.abs(ctx.apply_xt(B) - T_ref) / np.max(np.abs(T_ref), axis=0)) <= 1e-11
            Z_ref = X @ T_ref
            assert np.max(np.abs(ctx.apply_xxt(B) - Z_ref) / np.max(np.abs(Z_ref), axis=0)) <= 1e-11
        Tin = np.random.default_rng(4).standard_normal((ctx.P, 32))
        Y_ref = X @ Tin
        assert np.max(np.abs(ctx.apply_x(Tin) - Y_ref) / np.max(np.abs(Y_ref), axis=0)) <= 1e-11


def test_randomised_parity_sweep(built_lib):
    """scripts/fuzz_parity.py: 30 random (N, P, b, S, missing profile -- uniform / concentrated / log-normal per SNP --, forced or
    automatic missing-indicator route, with or without the sample-major copy) cases, exact-integer, fp64 and (one in four) fp32
    kernels against dense numpy (all-missing and monomorphic SNPs mixed in)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("FPCA_I8_MODE", "FPCA_DEBUG_I8_NOCOPY")}
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "30", "7"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all 30 cases ok" in r.stdout


@pytest.mark.parametrize("b,nq", [(16, 1), (16, 12), (16, 24), (32, 1), (32, 12), (32, 24), (64, 1), (64, 12)])
def test_k4_gram_and_block_gemm_at_full_height(fp, b, nq):
    """K4 directly, not through the solver: HipBackend::gram (k_gram + the plane-parallel split-K reduction,
    k_reduce_tall) and HipBackend::gemm (k_block_gemm) at N = 500,000 rows with 1, 12 and 24 basis blocks -- the basis at its
    16-column cap is 24 blocks -- against numpy.  N is not a multiple of the 512-row padding on purpose."""
    import ctypes as C

    N = 500000 - 77
    rng = np.random.default_rng(1000 * b + nq)
    V = np.asfortranarray(rng.standard_normal((N, nq * b)))
    W = np.asfortranarray(rng.standard_normal((N, b)))
    Cin = rng.standard_normal((nq, b, b))
    Cg = np.empty((nq, b, b))
    Out = np.empty((N, b), order="F")
    G = np.empty((b, b))
    with fp.Context.synthetic(N, 256, n_pop=4, accum="fp64") as ctx:
        for use_init, fused in ((1, 0), (0, 0), (1, 1)):  # fused: the update and the Gram matrix of its output from one launch
            fp._lib.check(fp.lib().fpca_debug_k4(ctx.h, b, nq, V.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
                                                 Cg.ctypes.data_as(C.c_void_p), Cin.ctypes.data_as(C.c_void_p), use_init,
                                                 Out.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p) if fused else None))
            ref = V @ Cin.reshape(nq * b, b) + (W if use_init else 0.0)
            assert np.max(np.abs(Out - ref)) <= 1e-12 * np.max(np.abs(ref)), (use_init,)
            if fused:
                Gr = ref.T @ ref
                assert np.max(np.abs(G - Gr)) <= 1e-12 * np.max(np.abs(Gr)), np.max(np.abs(G - Gr))
    Gref = (V.T @ W).reshape(nq, b, b)
    # sums of 500,000 products of O(1) numbers: |error| ~ eps sqrt(N) per entry at worst for a blocked summation
    assert np.max(np.abs(Cg - Gref)) <= 2e-13 * np.sqrt(N), np.max(np.abs(Cg - Gref))


@pytest.mark.parametrize("nq", [1, 3, 4, 13, 24, 27])  # (27: the basis at its cap + the waiting blocks; nq % 4 decides which wave takes Out'Out)
def test_k4_fused_update_and_gram_at_full_height(fp, nq):
    """Round 6: HipBackend::gemm_gramvw -- the update of the first Gram-Schmidt projection and the Gram matrices of the second from ONE
    pass over the basis (k_update_gram16: the four waves of a workgroup share a row tile and split the basis blocks) -- in place, as the
    solver calls it, at N = 500,000 - 77 rows against numpy; 32 columns take the two launches it replaces (same entry point)."""
    import ctypes as C

    N = 500000 - 77
    for b in ((16, 32) if nq == 3 else (16,)):
        rng = np.random.default_rng(77 * b + nq)
        V = np.asfortranarray(rng.standard_normal((N, nq * b)))
        W = np.asfortranarray(rng.standard_normal((N, b)))
        Cin = rng.standard_normal((nq, b, b)) * 1e-3
        Cg = np.empty((nq + 1, b, b))
        Out = np.empty((N, b), order="F")
        with fp.Context.synthetic(N, 256, n_pop=4, accum="fp64") as ctx:
            fp._lib.check(fp.lib().fpca_debug_k4_fused(ctx.h, b, nq, V.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
                                                       Cin.ctypes.data_as(C.c_void_p), Out.ctypes.data_as(C.c_void_p), Cg.ctypes.data_as(C.c_void_p)))
        ref = V @ Cin.reshape(nq * b, b) + W
        assert np.max(np.abs(Out - ref)) <= 1e-12 * np.max(np.abs(ref))
        Gref = np.concatenate([(V.T @ ref).reshape(nq, b, b), (ref.T @ ref)[None]], axis=0)
        # V' Out: sums of N products of O(1) numbers (as in the test above); Out' Out: its diagonal is O(N)
        assert np.max(np.abs(Cg[:nq] - Gref[:nq])) <= 2e-13 * np.sqrt(N) * max(1.0, np.max(np.abs(ref))), np.max(np.abs(Cg[:nq] - Gref[:nq]))
        assert np.max(np.abs(Cg[nq] - Gref[nq])) <= 1e-13 * np.max(np.abs(Gref[nq])), np.max(np.abs(Cg[nq] - Gref[nq]))


def test_thick_restart_at_full_height_keeps_an_orthonormal_basis(fp):
    """A forced thick restart (basis cap of 4 blocks) at N = 500,000: the compressed basis (Ritz rotation through
    k_block_gemm) must stay orthonormal and give the eigenpairs of the unrestricted solve; judged by a dense U'U on the host
    and by the reference's --check quantity."""
    N, P, k = 500000, 4000, 20
    with fp.Context.synthetic(N, P, n_pop=8, accum="auto") as ctx:  # 7 structured eigenvalues: k = 20 reaches into the bulk
        free = ctx.pca(ndim=k, tol=1e-7)
        r = ctx.pca(ndim=k, tol=1e-7, max_blocks=4)
        assert r["info"]["converged"] == 1 and r["info"]["restarts"] >= 2 and free["info"]["converged"] == 1
        assert np.max(np.abs(r["U"].T @ r["U"] - np.eye(k))) < 1e-10
        assert np.max(np.abs(r["d"] - free["d"]) / free["d"]) < 1e-9
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1.01e-7 * r["d"])


@pytest.mark.parametrize("nch,k", [(1, 8), (2, 8), (4, 8), (2, 20)])  # k = 20: two blocks of Ritz vectors to gather
def test_row_sharded_solver_path_on_one_rank(fp, monkeypatch, nch, k):
    """The row-sharded solver (backend.hpp RowShard; the default with several ranks) forced on ONE rank: all-gather /
    reduce-scatter go through RCCL's own ncclAllGather / ncclReduceScatter on the hardware (one rank: the only way this box
    can execute them), K3 runs in row chunks with the reduce-scatter of each chunk on the communication stream, the blocks
    are slices with the padded chunk layout -- and the solve must give what the plain path gives.  N is chosen so that the
    last chunk is cut by the end of the matrix."""
    N, P = 40000, 1500
    kb = -(-k // 16)
    with fp.Context.synthetic(N, P, n_pop=6, accum="i8") as ref:
        r0 = ref.pca(ndim=k, do_loadings=True)
    monkeypatch.setenv("FPCA_AR_CHUNKS", str(nch))
    monkeypatch.setenv("FPCA_FORCE_ROWSHARD", "1")
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        assert c.allreduce_chunks() == nch
        calls0, bytes0 = c.collective_stats()
        r = c.pca(ndim=k, do_loadings=True)
        calls, nbytes = c.collective_stats()
        nbytes -= bytes0
        # (the two paths sum the Gram matrices in different row orders; at k = 20 the 45-pass solve crosses its cheap-pass threshold
        #  within 0.6 % of it -- 8.045e-7 against 8.0e-7 at pass 44 -- so a rounding-level difference may end it one pass earlier)
        assert r["info"]["converged"] == 1 and abs(r["info"]["block_applies"] - r0["info"]["block_applies"]) <= (1 if k == 20 else 0)
        # per apply: nch all-gathers (of the operand's BYTE SLICES since round 6: S bytes per entry instead of 8) + one all-gather of
        # the b column maxima per rank + nch reduce-scatters; + nch all-gathers each for the download and for the loadings block;
        # + the scalar all-reduce of the trace (one rank: the Gram sums stay local)
        A, Ac = r["info"]["block_applies"], r["info"]["cheap_applies"]
        assert calls - calls0 == 2 * nch * A + Ac + 2 * nch * kb + 1
        full = nch * (((c.block_rows() + nch - 1) // nch + 511) // 512 * 512)  # rows of a whole block in the padded chunk layout
        # exact passes: the fp64 block (8 bytes per entry); cheap passes: 512 bytes of column maxima + cheap_slices bytes per entry
        assert nbytes == A * full * 16 * 8 + (A - Ac) * full * 16 * 8 + Ac * (64 * 8 + full * 16 * r["info"]["cheap_slices"]) + 2 * kb * full * 16 * 8 + 8
        assert Ac == 0 or nbytes < (A + 2 * kb) * full * 16 * 16  # (rounds 1-5: 8 + 8 bytes per entry and apply)
        assert np.max(np.abs(r["d"] - r0["d"]) / r0["d"]) < (1e-12 if r["info"]["block_applies"] == r0["info"]["block_applies"] else 1e-10)
        sg = np.sign(np.sum(r["U"] * r0["U"], axis=0))
        # the five structured pairs (6 sub-populations) are isolated: same vectors to rounding; the bulk pairs behind them are
        # only determined to tol x theta / gap by either solve (different row order = different summation order in the Grams)
        assert np.max(np.abs(r["U"][:, :5] * sg[:5] - r0["U"][:, :5])) < 1e-9 and np.max(np.abs(r["V"][:, :5] * sg[:5] - r0["V"][:, :5])) < 1e-9
        assert np.max(np.abs(r["U"] * sg - r0["U"])) < 1e-5
        assert np.max(np.abs(r["U"].T @ r["U"] - np.eye(k))) < 1e-10
        err, mse, rmse = c.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1.01e-6 * r["d"])
        assert np.max(np.abs(r["Px"] - r["U"] * np.sqrt(r["d"]))) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("ndim", [70, 130])  # 32- and 64-column blocks
def test_row_sharded_exchange_in_slices_at_wider_blocks(fp, monkeypatch, ndim):
    """The sliced exchange (k_slice_rows / k_unpack_slices) at the block widths the solver takes for more than 64 components: every pass
    in slices (test switch), sparse missing-call route (its fp64 operand comes out of the unpack pass), against the plain solve."""
    N, P = 20000, 1200
    with fp.Context.synthetic(N, P, n_pop=6, accum="i8") as ref:
        r0 = ref.pca(ndim=ndim, mixed=-1)
    monkeypatch.setenv("FPCA_AR_CHUNKS", "2")
    monkeypatch.setenv("FPCA_FORCE_ROWSHARD", "1")
    monkeypatch.setenv("FPCA_EXCHANGE_SLICES", "all")
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        r = c.pca(ndim=ndim, mixed=-1)
        assert r["info"]["solver_path"] == 1 and r["info"]["converged"] == 1 and r["info"]["blockvec"] == (32 if ndim == 70 else 64)
        assert np.max(np.abs(r["d"] - r0["d"]) / r0["d"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fallback_rank", "fp64_exchange"])
def test_row_sharded_exchange_format_does_not_depend_on_what_fitted_on_a_rank(fp, monkeypatch, case):
    """The row-sharded apply all-gathers BYTE SLICES of the block (round 6; by default in the passes on <= 4 slices, here in all).  The format is decided from the requested arithmetic and the
    transport alone, so that every rank issues the same collectives: a rank whose int8 buffers did not fit (FPCA_ACCUM_AUTO falls back
    to the fp64 kernels -- forced here) still sends slices of its rows, receives everybody's, and multiplies the block they spell with
    the fp64 kernels; FPCA_EXCHANGE_FP64 (test build) is rounds 1-5's exchange of the fp64 block.  Same eigenvalues either way."""
    N, P, k = 40000, 1500, 8
    with fp.Context.synthetic(N, P, n_pop=6, accum="auto") as ref:
        r0 = ref.pca(ndim=k)
    monkeypatch.setenv("FPCA_AR_CHUNKS", "2")
    monkeypatch.setenv("FPCA_FORCE_ROWSHARD", "1")
    monkeypatch.setenv("FPCA_DEBUG_I8_NOMEM" if case == "fallback_rank" else "FPCA_EXCHANGE_FP64", "1")
    monkeypatch.setenv("FPCA_EXCHANGE_SLICES", "all")  # (the exact passes too: the solve below makes no cheap ones)
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="auto") as c:
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        _, bytes0 = c.collective_stats()
        r = c.pca(ndim=k, mixed=-1)
        _, nbytes = c.collective_stats()
        A = r["info"]["block_applies"]
        full = 2 * (((c.block_rows() + 1) // 2 + 511) // 512 * 512)
        assert r["info"]["solver_path"] == 1 and r["info"]["converged"] == 1
        assert c.accum == ("fp64" if case == "fallback_rank" else "i8x7")
        per_entry = 7 + 8 if case == "fallback_rank" else 8 + 8  # all-gather + reduce-scatter bytes per entry of the block and apply
        assert A * full * 16 * per_entry <= nbytes - bytes0 <= A * full * 16 * per_entry + (A + 4) * 1024 + 2 * full * 16 * 8
        assert np.max(np.abs(r["d"] - r0["d"]) / r0["d"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("fault", ["selftest", "reduce_scatter"])
def test_row_sharded_exchange_failure_demotes_to_the_replicated_solver(fp, monkeypatch, fault):
    """The never-run default of a multi-GPU solve must not be a single point of failure: when the self-test
    of the row-sharded exchange fails, or a reduce-scatter of the solve reports an error half-way, the ranks agree -- one plain
    all-reduce of a flag through the same communicator -- and the solve ends on the replicated solver (one all-reduce of the
    N x b product per apply, svdwide.cpp:48-62 summed over ranks) with the same eigenvalues; fpca_pca_info.solver_path says
    which path ran, and the context does not try the failed layout again.  One rank over RCCL (this box has one GPU): the
    sharded code path is forced, the failure injected (test build only)."""
    N, P, k = 40000, 1500, 10
    with fp.Context.synthetic(N, P, n_pop=6, accum="i8") as ref:
        r0 = ref.pca(ndim=k, do_loadings=True)
        assert r0["info"]["solver_path"] == 0
    monkeypatch.setenv("FPCA_AR_CHUNKS", "2")
    monkeypatch.setenv("FPCA_FORCE_ROWSHARD", "1")
    if fault == "selftest":
        monkeypatch.setenv("FPCA_DEBUG_SELFTEST_FAIL", "all")
    else:
        monkeypatch.setenv("FPCA_DEBUG_RS_FAIL", "9")  # (two chunks per apply: the fifth apply fails)
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        r = c.pca(ndim=k, do_loadings=True)
        assert r["info"]["solver_path"] == (3 if fault == "selftest" else 4) and r["info"]["converged"] == 1
        assert r["info"]["block_applies"] == r0["info"]["block_applies"]  # whole blocks on one rank: the plain iteration
        assert np.max(np.abs(r["d"] - r0["d"]) / r0["d"]) < 1e-12
        sg = np.sign(np.sum(r["U"] * r0["U"], axis=0))
        assert np.max(np.abs(r["U"][:, :5] * sg[:5] - r0["U"][:, :5])) < 1e-9 and np.max(np.abs(r["V"][:, :5] * sg[:5] - r0["V"][:, :5])) < 1e-9
        monkeypatch.delenv("FPCA_DEBUG_SELFTEST_FAIL", raising=False)
        monkeypatch.delenv("FPCA_DEBUG_RS_FAIL", raising=False)
        r2 = c.pca(ndim=k)  # the failed layout is remembered: straight to the replicated solver, no second self-test
        assert r2["info"]["solver_path"] == (3 if fault == "selftest" else 4) and np.max(np.abs(r2["d"] - r0["d"]) / r0["d"]) < 1e-12
    if fault == "reduce_scatter":
        # ... and when the ranks CANNOT agree (the collective failed on this rank only, its peers never reach the agreement
        # -- here: the stream of the agreement's all-reduce looks pending for ever) the step is time-bounded: FPCA_ECOMM after the
        # limit instead of a hang, the communicator aborted, and every later collective of the context refused at once
        import time

        monkeypatch.setenv("FPCA_DEBUG_RS_FAIL", "9")
        monkeypatch.setenv("FPCA_DEBUG_AGREE_STALL", "1")
        monkeypatch.setenv("FPCA_AGREE_TIMEOUT_S", "1.5")
        with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:
            c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
            t0 = time.time()
            with pytest.raises(Exception, match="did not agree within|never completed"):
                c.pca(ndim=k)
            assert 1.0 < time.time() - t0 < 30
            monkeypatch.delenv("FPCA_DEBUG_RS_FAIL")
            monkeypatch.delenv("FPCA_DEBUG_AGREE_STALL")
            with pytest.raises(Exception, match="abandoned"):
                c.apply_xxt(np.zeros((N, 16)))
            # a new transport revives the context (fpca_comm_init_rank resets what was known about the old one)
            c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
            r3 = c.pca(ndim=k)
            assert r3["info"]["solver_path"] == 1 and np.max(np.abs(r3["d"] - r0["d"]) / r0["d"]) < 1e-10
    with fp.test_hooks(), fp.Context.synthetic(N, P, n_pop=6, accum="i8") as c:  # nothing injected: the sharded path itself
        c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        r = c.pca(ndim=k, partial_rows=True)
        assert r["info"]["solver_path"] == 1
        # one rank owns every row: the chunk-interleaved slice layout must put each row where it belongs
        assert r["row_ranges"] and sum(b - a for a, b in r["row_ranges"]) == N and not np.isnan(r["U"]).any() and not np.isnan(r["Px"]).any()
        sg = np.sign(np.sum(r["U"] * r0["U"], axis=0))
        assert np.max(np.abs(r["d"] - r0["d"]) / r0["d"]) < 1e-10 and np.max(np.abs(r["U"][:, :5] * sg[:5] - r0["U"][:, :5])) < 1e-9
        assert np.max(np.abs(r["Px"] - r["U"] * np.sqrt(r["d"]))) < 1e-12
