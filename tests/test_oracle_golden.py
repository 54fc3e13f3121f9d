"""CPU tests: the oracle (oracle/fpca_oracle.c, restated reference path) against the committed goldens
(tests/golden/golden_*.json, independent numpy dense eigh) and against the reference's documented semantics."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

CASES = [("data_chr1", 50), ("hapmap3_data", 10)]


def _open(golden_dir, name, stand):
    N = O.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    d = O.OracleData(os.path.join(golden_dir, name + ".bed"), N, stand)
    g = json.load(open(os.path.join(golden_dir, "golden_%s_%s.json" % (name, stand))))
    return d, g


def test_decode_tables():
    """data.cpp:65-148: bit order and code -> dosage map; header comment data.cpp:41-47."""
    byte = np.array([0b11100100], dtype=np.uint8)  # fields (LSB first): 00, 01, 10, 11
    out = np.zeros(4, dtype=np.uint8)
    O.lib().orc_decode_plink_simple(out.ctypes.data, byte.ctypes.data, 1)
    assert out.tolist() == [0, 1, 2, 3]
    O.lib().orc_decode_plink(out.ctypes.data, byte.ctypes.data, 1)
    assert out.tolist() == [2, 3, 1, 0]  # 00 -> 2, 01 -> NA(3), 10 -> 1, 11 -> 0


@pytest.mark.parametrize("name,k", CASES)
@pytest.mark.parametrize("stand", ["binom2", "binom"])
def test_sizes_stats_and_operator(golden_dir, name, k, stand):
    d, g = _open(golden_dir, name, stand)
    assert (d.N, d.P) == (g["N"], g["P"])  # nsnps from the file size (data.cpp:165-170)
    X = d.dense()
    ms = d.meansd()
    assert np.allclose(ms[:8, 0], g["mean_first8"], rtol=0, atol=0)
    assert np.allclose(ms[:8, 1], g["sd_first8"], rtol=1e-15, atol=0)
    assert abs(np.nansum(ms[:, 0]) - g["mean_sum"]) < 1e-9
    assert abs(np.nansum(ms[:, 1]) - g["sd_sum"]) < 1e-9
    assert abs((X * X).sum() - g["trace_raw"]) <= 1e-12 * g["trace_raw"]
    probe = np.cos(0.37 * np.arange(d.N) + 0.11) + 0.25
    for bs in (d.P, 400):  # one block, and block-streamed like --memory would (svdwide.h:57-68)
        op = O.OracleOp(d, bs)
        y = op.perform_op(probe)
        assert np.allclose(y[:8], g["probe_y_first8"], rtol=1e-11, atol=0)
        assert abs(np.linalg.norm(y) - g["probe_y_norm"]) <= 1e-12 * g["probe_y_norm"]
        assert abs(op.trace - g["trace_raw"]) <= 1e-12 * g["trace_raw"]  # accumulated on the first op only
        t = op.crossprod(probe)
        assert np.allclose(t, X.T @ probe, rtol=1e-11, atol=1e-9)
        assert np.allclose(op.prod(t), y, rtol=1e-11, atol=1e-6)


@pytest.mark.parametrize("name,k", CASES)
def test_reference_path_matches_golden(golden_dir, name, k):
    """RandomPCA::pca_fast with the CLI defaults (tol 1e-6, ncv = 2k+1) reproduces the dense eigendecomposition."""
    d, g = _open(golden_dir, name, "binom2")
    r = O.pca_fast(d, k, do_loadings=True)
    ev = np.array(g["eigenvalues_div_p"])[:k]
    assert np.max(np.abs(r["d"] - ev) / ev) < 1e-10
    assert abs(r["trace"] * d.P - g["trace_raw"]) <= 1e-12 * g["trace_raw"]
    assert np.max(np.abs(r["pve"] - np.array(g["pve"])[:k])) < 1e-12
    U5 = np.array(g["U_first5"]).T
    for c in range(5):
        assert abs(abs(U5[:, c] @ r["U"][:, c]) - 1) < 1e-10
    assert np.allclose(r["Px"], r["U"] * np.sqrt(r["d"]), rtol=1e-14)
    assert np.max(np.abs((r["V"] ** 2).sum(axis=0) - 1)) < 1e-6
    # iteration counts are in the range the survey measured with ARPACK (58 / 245 ops)
    assert 30 <= r["nops"] <= 400
    # the reference's --check quantity (randompca.cpp:663-703), README.md:207: "< 1e-8"
    err, mse, rmse = O.check(d, r["U"], r["d"], block_size=500)
    assert mse < 1e-8


def test_divisors(golden_dir):
    d, g = _open(golden_dir, "data_chr1", "binom2")
    for div, key in (("p", "eigenvalues_div_p"), ("n1", "eigenvalues_div_n1"), ("none", "eigenvalues_raw")):
        r = O.pca_fast(d, 5, div=div)
        ev = np.array(g[key])[:5]
        assert np.max(np.abs(r["d"] - ev) / ev) < 1e-10


def test_block_size_heuristic_and_format():
    # flashpca.cpp:636-686 with the default --memory 2048: 465 SNPs per block at 500k x 100k
    assert O.lib().orc_default_block_size(500000, 100000, 20, 0, 2048) == 465
    assert O.lib().orc_default_block_size(957, 14389, 10, 0, 2048) == 14389
    assert O.lib().orc_default_block_size(500000, 100000, 20, 0, 1) == 0
    # util.h:77: setprecision(7) default-float
    assert O.format_number(26.467988137205) == "26.46799"
    assert O.format_number(2.31179038) == "2.31179"
    assert O.format_number(1e-5) == "1e-05"
    assert O.format_number(0.5, 20) == "0.5"


def test_edge_cases_monomorphic_missing_and_padding():
    """data.cpp:299-320: sd <= 1e-9 -> zero column; all-missing -> NaN mean, zero column; pad bits ignored."""
    N, P = 10, 4
    npk = 3
    packed = np.zeros((P, npk), dtype=np.uint8)
    packed[0] = 0xFF  # all 11 -> dosage 0 everywhere -> sd 0
    packed[1] = 0x55  # all missing
    packed[2] = [0b11100100, 0b00100111, 0b00001011]  # mixed; last byte: 2 valid samples then pad bits
    packed[3] = packed[2]
    packed[3, 2] |= 0b11110000  # only the pad bits differ from SNP 2
    d = O.OracleData(packed=packed, N=N, P=P, stand="binom2")
    X = d.dense()
    ms = d.meansd()
    assert np.all(X[:, 0] == 0) and ms[0, 1] == 0
    assert np.all(X[:, 1] == 0) and np.isnan(ms[1, 0])
    assert np.array_equal(X[:, 2], X[:, 3])
    assert X[1, 2] == 0.0  # sample 1 of SNP 2 is missing (code 01) -> imputed to the mean -> 0
    lut = d.lookup()
    assert lut[2, 1] == 0.0 and lut[2, 3] < lut[2, 2] < lut[2, 0]


# Known answers printed in SURVEY.md section 8(c): computed by the surveyor's own throw-away numpy restatement, i.e. a
# third implementation independent of both oracle/ and tests/golden/make_golden.py.
SURVEY_KAT = {
    "hapmap3_data": dict(
        N=957, P=14389, code_counts={0: 2845704, 1: 21221, 2: 6131675, 3: 4771673}, snps_with_na=7823, monomorphic=0,
        eig=[26.467988137205, 23.514010331154, 6.957985524321, 6.085561261752, 4.425828379855, 2.664165852825,
             2.466606768219, 2.311790389987, 2.265891967763, 2.218548237411],
        trace_over_p=990.429613283002, pve1=0.026723744709,
        eigenvalues_txt=["26.46799", "23.51401", "6.957986", "6.085561", "4.425828", "2.664166", "2.466607", "2.31179",
                         "2.265892", "2.218548"]),
    "hm3_thinned": dict(N=957, P=14079,
                        eig=[26.160628146661, 23.393816760033, 6.915728456122, 6.049831829496, 4.402504579185, 2.651956041155,
                             2.466051075037, 2.309857529632, 2.264242412552, 2.21310683549],
                        trace_over_p=987.3561801334836),
    "data_chr1": dict(N=957, P=1129, eig=[28.011938222135, 25.068103599102, 7.805220828638, 6.847117690206, 5.00003151225],
                      eig50=2.877077388029, trace_over_p=987.2553072389829),
}


@pytest.mark.parametrize("name", sorted(SURVEY_KAT))
def test_oracle_reproduces_survey_known_answers(golden_dir, name):
    kat = SURVEY_KAT[name]
    d = O.OracleData(os.path.join(golden_dir, name + ".bed"), O.count_fam_rows(os.path.join(golden_dir, name + ".fam")), "binom2")
    assert (d.N, d.P) == (kat["N"], kat["P"])
    k = 50 if "eig50" in kat else 10
    r = O.pca_fast(d, k)
    n = len(kat["eig"])
    assert np.max(np.abs(r["d"][:n] - kat["eig"]) / np.array(kat["eig"])) < 1e-10
    assert abs(r["trace"] / 1.0 - kat["trace_over_p"]) < 1e-9 * kat["trace_over_p"]  # trace is already divided by P
    if "eig50" in kat:
        assert abs(r["d"][49] - kat["eig50"]) < 1e-10 * kat["eig50"]
    if "pve1" in kat:
        assert abs(r["pve"][0] - kat["pve1"]) < 1e-11
        assert [O.format_number(v) for v in r["d"]] == kat["eigenvalues_txt"]
    if "code_counts" in kat:
        raw = np.fromfile(os.path.join(golden_dir, name + ".bed"), dtype=np.uint8)
        assert raw[:3].tolist() == [0x6C, 0x1B, 0x01]
        rec = raw[3:].reshape(d.P, -1)
        codes = np.stack([(rec >> (2 * s)) & 3 for s in range(4)], axis=-1).reshape(d.P, -1)[:, :d.N]
        assert {c: int((codes == c).sum()) for c in range(4)} == kat["code_counts"]
        assert int(((codes == 1).sum(axis=1) > 0).sum()) == kat["snps_with_na"]
        ms = d.meansd() if np.isfinite(d.meansd()).all() else None
        d.dense()
        sd = d.meansd()[:, 1]
        assert int((~(sd > 1e-9)).sum()) == kat["monomorphic"]
