"""f-4, the input pipeline at size: a .bed of several 64 MB upload chunks read by 16 pread threads into the
two pinned bounce buffers, copied and re-pitched on the device -- as one shard and as three shards with snp_begin != 0 --
must land in HBM bit for bit (download_packed == the file body) with bit-equal per-SNP statistics; and the three header
bytes are validated (the reference seeks past them unseen, data.cpp:218)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fp(built_lib):
    import flashpca_amd

    return flashpca_amd


@pytest.fixture(scope="module")
def big_bed(fp, tmp_path_factory):
    """~320 MB: 40,001 samples (np = 10,001 bytes: NOT a multiple of the 128-byte device pitch, 1 valid + 3 pad samples in
    the last byte) x 32,000 SNPs = 5 upload chunks of 6,710 records."""
    N, P = 40001, 32000
    d = tmp_path_factory.mktemp("bigbed")
    path = str(d / "big.bed")
    with fp.Context.synthetic(N, P, n_pop=8) as ctx:
        packed = ctx.download_packed()
        ms, tr = ctx.stats()
    assert packed.size == ((N + 3) // 4) * P >= 300 * 10 ** 6
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(packed.tobytes())
    return dict(path=path, N=N, P=P, packed=packed, meansd=ms, trace=tr, dir=str(d))


@pytest.mark.parametrize("threads", [None, "3"])
def test_multi_chunk_upload_is_bit_exact(fp, big_bed, threads, monkeypatch):
    if threads:
        monkeypatch.setenv("FPCA_READ_THREADS", threads)
    N, P, np_ = big_bed["N"], big_bed["P"], (big_bed["N"] + 3) // 4
    assert (64 << 20) // np_ < P / 4  # at least five chunks
    with fp.Context.from_bed(big_bed["path"], N) as ctx:
        assert ctx.P == P and ctx.P_total == P
        assert np.array_equal(ctx.download_packed(), big_bed["packed"])
        ms, tr = ctx.stats()
        assert np.array_equal(ms, big_bed["meansd"]) and tr == big_bed["trace"]


def test_shards_with_offsets_are_bit_exact(fp, big_bed):
    """three shards [P r/3, P (r+1)/3): contiguous byte ranges 3 + np*begin (data.cpp:218), each several chunks"""
    N, P, np_ = big_bed["N"], big_bed["P"], (big_bed["N"] + 3) // 4
    body = big_bed["packed"].reshape(P, np_)
    tr = 0.0
    for r in range(3):
        lo, hi = P * r // 3, P * (r + 1) // 3
        with fp.Context.from_bed(big_bed["path"], N, snp_begin=lo, P=hi - lo) as sh:
            assert sh.P == hi - lo and sh.P_total == P
            assert np.array_equal(sh.download_packed().reshape(hi - lo, np_), body[lo:hi])
            ms, t = sh.stats()
            assert np.array_equal(ms, big_bed["meansd"][lo:hi])
            tr += t
    assert abs(tr - big_bed["trace"]) <= 1e-12 * big_bed["trace"]
    # a range past the end is refused
    with pytest.raises(fp.FpcaError):
        fp.Context.from_bed(big_bed["path"], N, snp_begin=P - 10, P=11)


def test_operator_on_uploaded_shards_adds_up(fp, big_bed):
    """X X' b summed over the three file shards == the one-shard operator (svdwide.cpp:48-62), default arithmetic"""
    N, P = big_bed["N"], big_bed["P"]
    b = np.random.default_rng(4).standard_normal((N, 2))
    with fp.Context.from_bed(big_bed["path"], N, accum="auto") as ctx:
        full = ctx.apply_xxt(b)
    acc = np.zeros_like(full)
    for r in range(3):
        lo, hi = P * r // 3, P * (r + 1) // 3
        with fp.Context.from_bed(big_bed["path"], N, snp_begin=lo, P=hi - lo, accum="auto") as sh:
            acc += sh.apply_xxt(b)
    assert np.max(np.abs(acc - full)) <= 1e-12 * np.max(np.abs(full))


@pytest.mark.parametrize("magic,what", [(b"\x6c\x1b\x00", "sample-major"), (b"\x00\x00\x00", "not a PLINK .bed"), (b"\x6c\x1b\x02", "unknown")])
def test_bed_magic_is_validated(fp, tmp_path, magic, what):
    N, P = 100, 40
    body = np.random.default_rng(1).integers(0, 256, size=P * 25, dtype=np.uint8).tobytes()
    path = str(tmp_path / "m.bed")
    open(path, "wb").write(magic + body)
    with pytest.raises(fp.FpcaError) as e:
        fp.Context.from_bed(path, N)
    assert what in str(e.value)
    open(path, "wb").write(b"\x6c\x1b\x01" + body)
    with fp.Context.from_bed(path, N) as ctx:
        assert ctx.P == P


def test_cli_rejects_sample_major_bed(fp, tmp_path, golden_dir):
    import shutil
    import subprocess

    for ext in (".bim", ".fam"):
        shutil.copy(os.path.join(golden_dir, "data_chr1" + ext), str(tmp_path / ("x" + ext)))
    raw = open(os.path.join(golden_dir, "data_chr1.bed"), "rb").read()
    open(str(tmp_path / "x.bed"), "wb").write(b"\x6c\x1b\x00" + raw[3:])
    r = subprocess.run([fp.CLI_PATH, "--bfile", str(tmp_path / "x"), "--notime"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "sample-major" in r.stderr
    assert not os.path.exists(str(tmp_path / "eigenvalues.txt"))
