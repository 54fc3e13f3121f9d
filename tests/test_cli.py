"""The flashpca drop-in CLI (flashpca_amd/csrc/cli_main.cpp).  CPU part: flag handling and exit codes, which mirror
flashpca.cpp:95-560.  GPU part: end-to-end run on the bundled fileset, outputs compared with the goldens."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "flashpca_amd", "_build", "flashpca")
DATA = os.path.join(ROOT, "tests", "golden", "data_chr1")


def run(args, cwd=None):
    return subprocess.run([CLI] + args, capture_output=True, text=True, cwd=cwd)


def test_cli_flag_errors(built_lib):
    r = run(["--version"])
    assert r.returncode == 0 and "flashpca 2.1" in r.stderr
    r = run(["--help"])
    assert r.returncode == 0 and "--bfile" in r.stderr and "--ndim" in r.stderr
    r = run(["--nosuchflag"])  # flashpca.cpp:100-106: parse errors exit with status 0
    assert r.returncode == 0 and "Use --help to get more help" in r.stderr
    r = run(["--notime"])
    assert r.returncode == 1 and "you must specify either --bfile or --bed / --fam / --bim" in r.stderr
    for extra, msg in ((["--ndim", "0"], "--ndim can't be less than 1"),
                       (["--standx", "sd"], "unknown standardization method (--standx): sd"),
                       (["--div", "q"], "unknown divisor (--div): q"),
                       (["--tol", "0"], "--tol can't be zero or negative"),
                       (["--maxiter", "0"], "--maxiter can't be less than 1"),
                       (["--precision", "1"], "output --precision too low"),
                       (["--memory", "10", "--blocksize", "5"], "cannot specify both --memory and --blocksize"),
                       (["--check", "--project"], "conflicting modes requested"),
                       (["--project"], "SNP-loadings must be specified using --inload"),
                       (["--accum", "bf16"], "unknown accumulate mode (--accum): bf16"),
                       (["--passes", "cheap"], "unknown --passes mode (mixed | exact): cheap"),
                       (["--scca"], "outside the PCA path")):
        r = run(["--bfile", DATA, "--notime"] + extra)
        assert r.returncode == 1, (extra, r.stderr)
        assert msg in r.stderr, (extra, r.stderr)
    r = run(["--bfile", "/nonexistent/prefix", "--notime"])
    assert r.returncode == 1 and "Exception: Error reading file" in r.stderr and "Terminating" in r.stderr
    assert r.stdout.startswith("arguments: flashpca ")


def test_cli_option_prefixes(built_lib):
    """po::parse_command_line (flashpca.cpp:97) runs with boost's default style, which includes allow_guessing: an
    unambiguous prefix of a long option selects it, the full name always wins, an ambiguous prefix is a parse error
    (reported like every parse error: message, "Use --help", exit status 0 -- flashpca.cpp:100-106)."""
    r = run(["--bf", DATA, "--notime", "--nd", "0"])  # --bfile, --ndim
    assert r.returncode == 1 and "--ndim can't be less than 1" in r.stderr
    r = run(["--bfile", DATA, "--not", "--outl", "l.txt", "--prec", "1"])  # --notime, --outload, --precision
    assert r.returncode == 1 and "output --precision too low" in r.stderr
    r = run(["--bfile", DATA, "--notime", "--outp", "x"])  # outpc, outpcx, outpcy, outpve, outproj
    assert r.returncode == 0 and "Use --help to get more help" in r.stderr
    assert "option '--outp' is ambiguous and matches '--outpc', '--outpcx', '--outpcy', '--outpve', and '--outproj'" in r.stderr
    r = run(["--bfile", DATA, "--notime", "--outpc", "x", "--nd", "0"])  # a full name is never ambiguous (outpc / outpcx / outpcy)
    assert r.returncode == 1 and "--ndim can't be less than 1" in r.stderr
    r = run(["--b", DATA])  # batch, blocksize, bed, bim, bfile
    assert r.returncode == 0 and "is ambiguous and matches '--batch', '--blocksize', '--bed', '--bim', and '--bfile'" in r.stderr
    r = run(["--stand", "binom"])
    assert r.returncode == 0 and "ambiguous and matches '--standx' and '--standy'" in r.stderr
    # options this build adds are matched by their full names only, so that the reference's abbreviations keep their
    # meaning: --de is the reference's --debug, --max its --maxiter (not --device / --maxblocks)
    r = run(["--bfile", DATA, "--notime", "--de", "--max", "0"])
    assert r.returncode == 1 and "--maxiter can't be less than 1" in r.stderr
    r = run(["--bfile", DATA, "--notime", "--dev", "0"])
    assert r.returncode == 0 and "unrecognised option '--dev'" in r.stderr
    r = run(["--bfile", DATA, "--notime", "--pass", "exact"])  # (--passes is this build's: full name only)
    assert r.returncode == 0 and "unrecognised option '--pass'" in r.stderr
    # CCA-only options of the reference are accepted (and ignored) like any other registered option
    r = run(["--bfile", DATA, "--notime", "--lambda1", "0.1", "--outpcx", "a", "--save-vinit", "--ndim", "0"])
    assert r.returncode == 1 and "--ndim can't be less than 1" in r.stderr


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path, built_lib):
    g = json.load(open(DATA.replace("data_chr1", "golden_data_chr1_binom2.json")))
    r = run(["--bfile", DATA, "--ndim", "10", "--outload", "loadings.txt", "--outmeansd", "meansd.txt", "--notime"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[1].startswith("Start flashpca (version 2.1")
    for must in ("PCA begin", "PCA done", "Writing 10 eigenvalues to file eigenvalues.txt",
                 "Writing 10 eigenvectors to file eigenvectors.txt", "Writing 10 PCs to file pcs.txt",
                 "Writing 10 proportion variance explained to file pve.txt", "Goodbye!"):
        assert must in lines, must
    ev = np.loadtxt(tmp_path / "eigenvalues.txt")
    assert np.allclose(ev, np.array(g["eigenvalues_div_p"])[:10], rtol=2e-6)  # 7 significant digits
    expect = ["%.7g" % v for v in g["eigenvalues_div_p"][:10]]
    assert open(tmp_path / "eigenvalues.txt").read().split() == expect
    assert open(tmp_path / "pve.txt").read().split() == ["%.7g" % v for v in g["pve"][:10]]
    vec = open(tmp_path / "eigenvectors.txt").read().splitlines()
    assert vec[0] == "FID\tIID\t" + "\t".join("U%d" % i for i in range(1, 11))
    assert len(vec) == 958 and vec[1].split("\t")[:2] == ["2431", "NA19916"]
    pcs = open(tmp_path / "pcs.txt").read().splitlines()
    assert pcs[0] == "FID\tIID\t" + "\t".join("PC%d" % i for i in range(1, 11))
    U = np.array([l.split("\t")[2:] for l in vec[1:]], dtype=float)
    PC = np.array([l.split("\t")[2:] for l in pcs[1:]], dtype=float)
    assert np.allclose(PC, U * np.sqrt(ev), rtol=1e-5, atol=1e-7)
    U5 = np.array(g["U_first5"]).T
    for c in range(5):
        assert abs(abs(U5[:, c] @ U[:, c]) - 1) < 1e-5
    load = open(tmp_path / "loadings.txt").read().splitlines()
    assert load[0] == "SNP\tRefAllele\t" + "\t".join("V%d" % i for i in range(1, 11)) and len(load) == 1130
    assert load[1].split("\t")[:2] == ["rs4970383", "A"]
    ms = open(tmp_path / "meansd.txt").read().splitlines()
    assert ms[0] == "SNP\tRefAllele\tMean\tSD" and len(ms) == 1130
    assert ms[1].split("\t")[2] == "%.7g" % g["mean_first8"][0]
    # --check re-reads eigenvectors/eigenvalues (randompca.cpp:627-661); 7-digit files -> mse ~1e-12
    # (printed under --verbose only, like the reference: randompca.cpp:670-700)
    r = run(["--bfile", DATA, "--check", "--notime"], cwd=tmp_path)
    assert r.returncode == 0 and "Mean squared error:" not in r.stdout
    r = run(["--bfile", DATA, "--check", "--notime", "--verbose"], cwd=tmp_path)
    assert r.returncode == 0 and "Mean squared error:" in r.stdout
    mse = float(r.stdout.split("Mean squared error: ")[1].split(",")[0])
    assert mse < 1e-8
    # --project with the just-written loadings and mean/sd reproduces the PCs (flashpcaR test_project.R:12-47, tol 1e-5)
    r = run(["--bfile", DATA, "--project", "--inload", "loadings.txt", "--inmeansd", "meansd.txt", "--outproj", "proj.txt", "--notime"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    pr = open(tmp_path / "proj.txt").read().splitlines()
    assert pr[0] == pcs[0]
    PR = np.array([l.split("\t")[2:] for l in pr[1:]], dtype=float)
    assert np.max(np.abs(PR - PC)) < 1e-4
    # mixed fp32 mode: same files, eigenvalues identical at the 7 printed digits
    r = run(["--bfile", DATA, "--ndim", "10", "--accum", "fp32", "--suffix", "_fp32.txt", "--notime"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    ev32 = np.loadtxt(tmp_path / "eigenvalues_fp32.txt")
    assert np.allclose(ev32, ev, rtol=1e-6)
    # exact-integer int8 mode: fp64-equivalent results
    r = run(["--bfile", DATA, "--ndim", "10", "--accum", "i8", "--suffix", "_i8.txt", "--notime"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert np.allclose(np.loadtxt(tmp_path / "eigenvalues_i8.txt"), ev, rtol=1e-6)
    r = run(["--bfile", DATA, "--ndim", "10", "--accum", "i9", "--notime"], cwd=tmp_path)
    assert r.returncode == 1 and "unknown accumulate mode" in r.stderr
    # ndim limit (flashpca.cpp:623-633)
    r = run(["--bfile", DATA, "--ndim", "500", "--notime"], cwd=tmp_path)
    assert r.returncode == 1 and "You asked for 500 dimensions, but only 478allowed" in r.stderr


@pytest.mark.gpu
def test_randomised_cli_sweep(built_lib):
    """scripts/fuzz_cli.py: 12 random filesets through the binary (PCA with loadings and mean/sd, --project of the same
    samples == the PCs, --check), every output file compared with numpy at the written precision."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_cli.py"), "12", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all 12 cases ok" in r.stdout


def _tab(path, skip=2):
    return np.array([l.split("\t")[skip:] for l in open(path).read().splitlines()[1:]], dtype=float)


@pytest.mark.gpu
def test_cli_gpus_launcher(tmp_path, built_lib, golden_dir):
    """flashpca --gpus G: one process per SNP shard, the parent is rank 0 and writes every file.  On a one-GPU box the
    ranks share the device and sum their partial products through host shared memory (FPCA_CLI_TEST_TRANSPORT=shm -- RCCL
    refuses two ranks on one device), which exercises the fork / rendezvous / sharding / gather logic; every output must
    equal the single-process run.  Without the test transport the run either works (enough GPUs: real RCCL) or ends with
    the out-of-range device message, never a hang."""
    import flashpca_amd as fp

    data = os.path.join(golden_dir, "hapmap3_data")
    base = ["--bfile", data, "--ndim", "10", "--outload", "load.txt", "--outmeansd", "ms.txt", "--precision", "12"]
    outs = {}
    for g, env_extra in ((1, {}), (2, {"FPCA_CLI_TEST_TRANSPORT": "shm"}), (3, {"FPCA_CLI_TEST_TRANSPORT": "shm"})):
        d = tmp_path / ("g%d" % g)
        d.mkdir()
        # (the host-memory transport exists only in the -DFPCA_TEST_HOOKS build of the CLI; g = 1 is the shipped binary)
        args = [fp.HOOKS_CLI_PATH if g > 1 else fp.CLI_PATH] + base + (["--gpus", str(g)] if g > 1 else [])
        r = subprocess.run(args, cwd=d, capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert r.stdout.count("Goodbye!") == 1 and r.stdout.count("PCA done") == 1  # only rank 0 talks
        outs[g] = d
    e1 = np.loadtxt(outs[1] / "eigenvalues.txt")
    U1, V1, m1 = _tab(outs[1] / "eigenvectors.txt"), _tab(outs[1] / "load.txt"), _tab(outs[1] / "ms.txt")
    for g in (2, 3):
        e, U, V, m = np.loadtxt(outs[g] / "eigenvalues.txt"), _tab(outs[g] / "eigenvectors.txt"), _tab(outs[g] / "load.txt"), _tab(outs[g] / "ms.txt")
        sg = np.sign(np.sum(U1 * U, axis=0))
        assert np.max(np.abs(e - e1) / e1) < 1e-10
        assert np.max(np.abs(U * sg - U1)) < 1e-9 and np.max(np.abs(V * sg - V1)) < 1e-9
        assert np.array_equal(m, m1)
        assert np.max(np.abs(_tab(outs[g] / "pcs.txt") * sg - _tab(outs[1] / "pcs.txt"))) < 1e-8
    d = tmp_path / "real"
    d.mkdir()
    r = subprocess.run([fp.CLI_PATH] + base + ["--gpus", "2"], cwd=d, capture_output=True, text=True, timeout=300)
    if r.returncode == 0:
        assert np.max(np.abs(np.loadtxt(d / "eigenvalues.txt") - e1) / e1) < 1e-10
    else:
        assert "device index out of range" in r.stderr
    r = subprocess.run([fp.CLI_PATH, "--bfile", data, "--check", "--gpus", "2"], cwd=d, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "--gpus applies to PCA only" in r.stderr


@pytest.mark.gpu
def test_cli_gpus_launcher_rccl_shaped_exchange(tmp_path, built_lib):
    """The row-sharded solver with a transport that HAS all-gather / reduce-scatter (FPCA_CLI_TEST_TRANSPORT=shm2 installs
    fpca_set_collectives over host shared memory): the very call sequence of the RCCL path -- per row chunk, the reduce-scatter
    of chunk i on the communication stream under the K3 of chunk i + 1, the chunk-interleaved slice layout, chunks wholly behind
    the last row skipped -- with 2, 3, 4 and 8 processes on one device and 1 / 2 / 4 row chunks (5,000 samples: 5,120 padded rows, so
    the chunks are whole, clipped and empty).  The exchange self-test (HipBackend::exchange_selftest) runs inside every one of
    these; every output must equal the single-process run."""
    import flashpca_amd as fp

    N, P, k = 5000, 3000, 8
    pre = str(tmp_path / "syn")
    with fp.Context.synthetic(N, P, n_pop=6, accum="fp64") as c:
        packed = c.download_packed()
    with open(pre + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        packed.tofile(f)
    open(pre + ".fam", "w").write("".join("F%d I%d 0 0 0 -9\n" % (i, i) for i in range(N)))
    open(pre + ".bim", "w").write("".join("1 rs%d 0 %d A C\n" % (j, j + 1) for j in range(P)))
    base = ["--bfile", pre, "--ndim", str(k), "--outload", "load.txt", "--outmeansd", "ms.txt", "--precision", "12"]
    d1 = tmp_path / "one"
    d1.mkdir()
    r = subprocess.run([fp.CLI_PATH] + base, cwd=d1, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    e1 = np.loadtxt(d1 / "eigenvalues.txt")
    U1, V1, m1 = _tab(d1 / "eigenvectors.txt"), _tab(d1 / "load.txt"), _tab(d1 / "ms.txt")
    # (8 processes = the shard plan of BASELINE configs[3] / [4]: the 5,120 padded rows in 1 / 2 / 4 chunks of 8 pieces each)
    for g, chunks in ((2, 1), (2, 4), (3, 2), (3, 4), (4, 4), (8, 1), (8, 2), (8, 4)):
        d = tmp_path / ("g%dc%d" % (g, chunks))
        d.mkdir()
        env = dict(os.environ, FPCA_CLI_TEST_TRANSPORT="shm2", FPCA_AR_CHUNKS=str(chunks))
        r = subprocess.run([fp.HOOKS_CLI_PATH] + base + ["--gpus", str(g)], cwd=d, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (g, chunks, r.stdout[-1500:] + r.stderr[-1500:])
        e, U, V, m = np.loadtxt(d / "eigenvalues.txt"), _tab(d / "eigenvectors.txt"), _tab(d / "load.txt"), _tab(d / "ms.txt")
        sg = np.sign(np.sum(U1 * U, axis=0))
        assert np.max(np.abs(e - e1) / e1) < 1e-10, (g, chunks)
        assert np.max(np.abs(U * sg - U1)) < 1e-8 and np.max(np.abs(V * sg - V1)) < 1e-8, (g, chunks)
        assert np.array_equal(m, m1)


@pytest.mark.gpu
def test_cli_gpus_solver_switch_and_fail_safe_demotion(tmp_path, built_lib):
    """`--solver rowshard|replicated` and the fail-safe of the row-sharded default (1a), with 2 and 3
    processes on one device over the host-memory transports: (i) --solver replicated -- north_star's literal scheme, ONE
    all-reduce of the N x b product per pass -- gives the single-process files; (ii) a reduce-scatter that reports a failure
    half-way through the row-sharded solve (every rank: the call sequence is the same everywhere), a self-test that fails on
    ONE rank only, and one that fails everywhere all end on the replicated solver with the same eigenvalues (1e-12) and say so
    under --verbose; every rank has written its own rows of the eigenvectors / PCs into the shared region (1c: no funnel
    through rank 0), so the files are complete.  Unknown --solver values are refused like unknown --accum values."""
    import flashpca_amd as fp

    N, P, k = 5000, 3000, 8
    pre = str(tmp_path / "syn")
    with fp.Context.synthetic(N, P, n_pop=6, accum="fp64") as c:
        packed = c.download_packed()
    with open(pre + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        packed.tofile(f)
    open(pre + ".fam", "w").write("".join("F%d I%d 0 0 0 -9\n" % (i, i) for i in range(N)))
    open(pre + ".bim", "w").write("".join("1 rs%d 0 %d A C\n" % (j, j + 1) for j in range(P)))
    base = ["--bfile", pre, "--ndim", str(k), "--outload", "load.txt", "--outmeansd", "ms.txt", "--precision", "14", "--verbose"]
    d1 = tmp_path / "one"
    d1.mkdir()
    r = subprocess.run([fp.CLI_PATH] + base, cwd=d1, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    e1 = np.loadtxt(d1 / "eigenvalues.txt")
    U1, V1, m1 = _tab(d1 / "eigenvectors.txt"), _tab(d1 / "load.txt"), _tab(d1 / "ms.txt")
    cases = [
        ("replicated", 2, "shm", ["--solver", "replicated"], {}, "eigensolver layout over 2 GPUs: replicated\n"),
        ("replicated3", 3, "shm2", ["--solver", "replicated"], {}, "eigensolver layout over 3 GPUs: replicated\n"),
        ("rowshard", 3, "shm2", ["--solver", "rowshard"], {}, "eigensolver layout over 3 GPUs: row-sharded\n"),
        # (round 6: the all-gather of byte slices of the operand -- by default in the passes on <= 4 slices, here in every pass)
        ("rowshard_slices", 3, "shm2", ["--solver", "rowshard"], {"FPCA_EXCHANGE_SLICES": "all"}, "eigensolver layout over 3 GPUs: row-sharded\n"),
        ("rsfail", 2, "shm2", [], {"FPCA_DEBUG_RS_FAIL": "7", "FPCA_AR_CHUNKS": "2"}, "a collective of the row-sharded solve failed"),
        ("rsfail3", 3, "shm2", [], {"FPCA_DEBUG_RS_FAIL": "5"}, "a collective of the row-sharded solve failed"),
        ("selftest1", 3, "shm2", [], {"FPCA_DEBUG_SELFTEST_FAIL": "1"}, "the self-test of the row-sharded exchange failed"),
        ("selftestall", 2, "shm", [], {"FPCA_DEBUG_SELFTEST_FAIL": "all"}, "the self-test of the row-sharded exchange failed"),
    ]
    for name, g, transport, extra, env_extra, says in cases:
        d = tmp_path / name
        d.mkdir()
        env = dict(os.environ, FPCA_CLI_TEST_TRANSPORT=transport, **env_extra)
        r = subprocess.run([fp.HOOKS_CLI_PATH] + base + ["--gpus", str(g)] + extra, cwd=d, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (name, r.stdout[-1500:] + r.stderr[-1500:])
        assert says in r.stdout, (name, r.stdout[-1500:])
        if env_extra.get("FPCA_DEBUG_RS_FAIL") or env_extra.get("FPCA_DEBUG_SELFTEST_FAIL"):
            assert r.stderr.count("continues with the replicated solver") + r.stderr.count("starts over with the replicated solver") == g, (name, r.stderr[-1500:])
        e, U, V, m = np.loadtxt(d / "eigenvalues.txt"), _tab(d / "eigenvectors.txt"), _tab(d / "load.txt"), _tab(d / "ms.txt")
        sg = np.sign(np.sum(U1 * U, axis=0))
        # the replicated solver follows the one-GPU iteration (whole blocks, same start block): eigenvalues to rounding
        assert np.max(np.abs(e - e1) / e1) < (1e-10 if name.startswith("rowshard") else 1e-12), (name, np.max(np.abs(e - e1) / e1))
        assert np.max(np.abs(U * sg - U1)) < 1e-8 and np.max(np.abs(V * sg - V1)) < 1e-8, name
        assert np.max(np.abs(_tab(d / "pcs.txt") * sg - _tab(d1 / "pcs.txt"))) < 1e-7, name
        assert np.array_equal(m, m1)
    r = subprocess.run([fp.CLI_PATH] + base + ["--solver", "sharded"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "unknown --solver layout" in r.stderr


@pytest.mark.gpu
def test_cli_gpus_launcher_failures_do_not_hang(tmp_path, built_lib, golden_dir):
    """A rank that dies (SIGKILL: what an OOM kill looks like), or rank 0 failing after the fork, must end the whole --gpus
    run promptly with a message and a non-zero status; what can be refused from the file sizes is refused before the fork."""
    import time

    import flashpca_amd as fp

    data = os.path.join(golden_dir, "hapmap3_data")
    env = dict(os.environ, FPCA_CLI_TEST_TRANSPORT="shm")
    for victim, text in (("1", "died unexpectedly"), ("2", "died unexpectedly"), ("0", "injected failure of rank 0")):
        t0 = time.time()
        r = subprocess.run([fp.HOOKS_CLI_PATH, "--bfile", data, "--ndim", "5", "--gpus", "3"], cwd=tmp_path, capture_output=True, text=True,
                           env=dict(env, FPCA_CLI_TEST_KILL_RANK=victim), timeout=120)
        assert r.returncode == 1 and text in r.stderr, (victim, r.stdout[-800:], r.stderr[-800:])
        assert time.time() - t0 < 60
        assert not os.path.exists(tmp_path / "eigenvalues.txt")
    # A collective that fails on ONE rank only (forbidden by the hook contract of fpca.h): rank 1 leaves its 3rd
    # reduce-scatter for the agreement while ranks 0 and 2 are inside that reduce-scatter.  The transport sees ranks in different
    # calls and fails on all of them, the agreement cannot complete, every rank returns FPCA_ECOMM: the run ends -- promptly, with a
    # message and a non-zero status, no output files -- instead of hanging or, worse, summing unrelated buffers.
    t0 = time.time()
    r = subprocess.run([fp.HOOKS_CLI_PATH, "--bfile", data, "--ndim", "5", "--gpus", "3", "--solver", "rowshard"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, FPCA_CLI_TEST_TRANSPORT="shm2", FPCA_DEBUG_RS_FAIL="3", FPCA_DEBUG_RS_FAIL_RANK="1"), timeout=120)
    assert r.returncode == 1 and "not in the same collective" in r.stderr and "abandoned" in r.stderr, (r.stdout[-800:], r.stderr[-1500:])
    assert time.time() - t0 < 60 and not os.path.exists(tmp_path / "eigenvalues.txt")
    # refused before the fork: ndim limit, .bim / .bed mismatch when a file with SNP row names is asked for
    r = subprocess.run([fp.HOOKS_CLI_PATH, "--bfile", data, "--ndim", "500", "--gpus", "3"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 1 and "You asked for 500 dimensions, but only 478allowed" in r.stderr
    import shutil

    for ext in (".bed", ".fam"):
        shutil.copy(data + ext, str(tmp_path / ("mm" + ext)))
    lines = open(data + ".bim").read().splitlines(True)
    open(tmp_path / "mm.bim", "w").writelines(lines[:-7])
    for extra in ([], ["--gpus", "2"]):
        r = subprocess.run([fp.HOOKS_CLI_PATH, "--bfile", str(tmp_path / "mm"), "--ndim", "5", "--outload", "l.txt"] + extra, cwd=tmp_path,
                           capture_output=True, text=True, env=env, timeout=60)
        assert r.returncode == 1 and "different number of SNPs" in r.stderr
        assert not os.path.exists(tmp_path / "eigenvalues.txt")  # nothing was computed or written first
    # without a file that needs the names the mismatch does not matter (the reference never consults the .bim for sizes)
    r = subprocess.run([fp.CLI_PATH, "--bfile", str(tmp_path / "mm"), "--ndim", "5"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [100, 200])
def test_cli_more_components_than_the_block_width(tmp_path, built_lib, golden_dir, k):
    """flashpca --ndim 100 / 200 (the reference admits up to 478 here): files against a dense eigendecomposition."""
    import flashpca_amd as fp
    from oracle import oracle as O

    data = os.path.join(golden_dir, "hapmap3_data")
    r = subprocess.run([fp.CLI_PATH, "--bfile", data, "--ndim", str(k), "--precision", "14", "--notime"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    N = O.count_fam_rows(data + ".fam")
    od = O.OracleData(data + ".bed", N, "binom2")
    X = od.dense()
    w = np.linalg.eigvalsh(X @ X.T)[::-1][:k] / od.P
    e = np.loadtxt(tmp_path / "eigenvalues.txt")
    assert e.shape == (k,) and np.max(np.abs(e - w) / w) < 1e-9
    U = _tab(tmp_path / "eigenvectors.txt")
    assert U.shape == (N, k) and np.max(np.abs(U.T @ U - np.eye(k))) < 1e-9
    res = np.linalg.norm(X @ (X.T @ U) / od.P - U * e, axis=0)
    assert np.max(res / e) < 2e-6
    assert np.max(np.abs(_tab(tmp_path / "pcs.txt") - U * np.sqrt(e))) < 1e-10
