"""Randomised sweep of the process boundary (GPU): random PLINK filesets on disk -> the flashpca binary -> its text outputs
against numpy (eigenvalues, eigenvectors up to sign, PCs, pve, loadings, mean/sd), then --project and --check on the
same files.  Exercises N % 4 != 0, tabs / multiple spaces as separators, missing calls, --standx / --div / --precision,
and the --gpus launcher (2-4 ranks on one device through the test transport).  python scripts/fuzz_cli.py [cases] [seed]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CLI = fp.CLI_PATH


def table(path, skipcols=2):
    rows = [l.split("\t") for l in open(path).read().splitlines()[1:]]
    return np.array([r[skipcols:] for r in rows], dtype=float)


t0 = time.time()
for case in range(ncases):
    N = int(rng.integers(30, 700))
    P = int(rng.integers(40, 900))
    k = int(min((min(N, P) - 1) // 2, rng.choice([1, 2, 5, 10, 20])))
    stand = str(rng.choice(["binom2", "binom"]))
    div = str(rng.choice(["p", "n1", "none"]))
    prec = int(rng.choice([7, 10, 15]))
    npop = int(rng.integers(2, 8))
    pop = rng.integers(0, npop, size=N)
    f = np.clip(rng.uniform(0.1, 0.9, size=(P, 1)) + 0.2 * rng.standard_normal((P, npop)), 0.05, 0.95)
    g = rng.binomial(2, f[:, pop])
    codes = np.select([g == 2, g == 1], [0, 2], 3).astype(np.uint8)
    codes[rng.random(codes.shape) < float(rng.choice([0.0, 0.003, 0.03]))] = 1
    pad = (-N) % 4
    cp = np.concatenate([codes, np.zeros((P, pad), dtype=np.uint8)], axis=1) if pad else codes
    packed = (cp[:, 0::4] | (cp[:, 1::4] << 2) | (cp[:, 2::4] << 4) | (cp[:, 3::4] << 6)).astype(np.uint8)
    with tempfile.TemporaryDirectory() as td:
        pre = os.path.join(td, "d")
        with open(pre + ".bed", "wb") as fh:
            fh.write(bytes([0x6C, 0x1B, 0x01]))
            fh.write(packed.tobytes())
        sep = ["\t", " ", "  ", " \t "][int(rng.integers(4))]
        with open(pre + ".fam", "w") as fh:
            lines = [sep.join(["F%d" % i, "I%d" % i, "0", "0", "0", "-9"]) for i in range(N)]
            fh.write("\n".join(lines) + "\n")
        with open(pre + ".bim", "w") as fh:
            fh.write("".join(sep.join(["1", "rs%d" % j, "0", str(j + 1), "A", "C"]) + "\n" for j in range(P)))
        # numpy reference
        G = np.select([codes == 0, codes == 2, codes == 3], [2.0, 1.0, 0.0], np.nan).T
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = np.nansum(G, axis=0) / np.sum(~np.isnan(G), axis=0)
            pp = mean / 2
            sd = np.sqrt(2 * pp * (1 - pp)) if stand == "binom2" else np.sqrt(pp * (1 - pp))
            X = (G - mean) / sd
        X[:, ~(sd > 1e-9)] = 0.0
        X[np.isnan(X)] = 0.0
        dv = {"p": P, "n1": N - 1, "none": 1}[div]
        w, v = np.linalg.eigh(X @ X.T / dv)
        w, v = w[::-1], v[:, ::-1]
        args = [CLI, "--bfile", pre, "--ndim", str(k), "--standx", stand, "--div", div, "--precision", str(prec), "--outload", "load.txt",
                "--outmeansd", "ms.txt", "--tol", "1e-9"]
        G = int(rng.choice([1, 1, 2, 3, 4]))  # --gpus G through the host-shared-memory test transport (one GPU here)
        env = dict(os.environ)
        if G > 1:  # (that transport exists only in the -DFPCA_TEST_HOOKS build of the CLI)
            args = [fp.HOOKS_CLI_PATH] + args[1:] + ["--gpus", str(G)]
            env["FPCA_CLI_TEST_TRANSPORT"] = "shm"
        r = subprocess.run(args, cwd=td, capture_output=True, text=True, env=env)
        desc = dict(N=N, P=P, k=k, stand=stand, div=div, prec=prec, gpus=G)
        if r.returncode != 0:
            print("case", case, desc, "CLI FAILED", r.stdout[-500:], r.stderr[-500:])
            sys.exit(1)
        ev = np.loadtxt(os.path.join(td, "eigenvalues.txt"), ndmin=1)
        U = table(os.path.join(td, "eigenvectors.txt"))
        pcs = table(os.path.join(td, "pcs.txt"))
        pve = np.loadtxt(os.path.join(td, "pve.txt"), ndmin=1)
        V = table(os.path.join(td, "load.txt"))
        ms = table(os.path.join(td, "ms.txt"))
        rt = 10.0 ** (1 - prec) * 5
        e_val = float(np.max(np.abs(ev - w[:k]) / w[0]))
        e_res = float(np.max(np.linalg.norm(X @ (X.T @ U) / dv - U * ev, axis=0)) / w[0])
        e_pcs = float(np.max(np.abs(pcs - U * np.sqrt(ev))) / np.sqrt(w[0]))
        e_pve = float(np.max(np.abs(pve - ev / (np.sum(X * X) / dv))))
        good = np.isfinite(sd) & (sd > 1e-9)
        e_ms = float(max(np.max(np.abs(ms[good, 0] - mean[good])), np.max(np.abs(ms[good, 1] - sd[good]))))
        with np.errstate(invalid="ignore", divide="ignore"):
            Vref = X.T @ U / np.sqrt(ev) / np.sqrt(dv)
        e_v = float(np.max(np.abs(V - Vref)))
        # projection of the same samples from the written loadings / mean-sd == the PCs (HapMap3/test_pca.R:213-228)
        r2 = subprocess.run([CLI, "--bfile", pre, "--project", "--inload", "load.txt", "--inmeansd", "ms.txt", "--outproj", "proj.txt", "--div", div,
                             "--precision", str(prec)], cwd=td, capture_output=True, text=True)
        if r2.returncode != 0:
            print("case", case, desc, "PROJECT FAILED", r2.stdout[-500:], r2.stderr[-500:])
            sys.exit(1)
        proj = table(os.path.join(td, "proj.txt"))
        e_proj = float(np.max(np.abs(proj - pcs)) / np.sqrt(w[0]))
        r3 = subprocess.run([CLI, "--bfile", pre, "--check", "--verbose", "--outvec", "eigenvectors.txt", "--outval", "eigenvalues.txt", "--standx", stand, "--div", div],
                            cwd=td, capture_output=True, text=True)
        ok3 = r3.returncode == 0 and r3.stdout.count("eval") >= k
        tol_txt = max(rt, 1e-8)
        ok = e_val < tol_txt and e_res < 10 * tol_txt + 1e-7 and e_pcs < 10 * tol_txt and e_pve < tol_txt and e_ms < tol_txt * 3 and e_v < 20 * tol_txt \
            and e_proj < 50 * tol_txt and ok3 and U.shape == (N, k)
        if not ok or case % 5 == 0:
            print("case %3d %s eval %.1e resid %.1e pcs %.1e pve %.1e meansd %.1e load %.1e project %.1e check %s %s" % (
                case, desc, e_val, e_res, e_pcs, e_pve, e_ms, e_v, e_proj, ok3, "OK" if ok else "FAIL"), flush=True)
        if not ok:
            print(r3.stdout[-800:], r3.stderr[-300:])
            sys.exit(1)
print("all %d cases ok, %.0f s" % (ncases, time.time() - t0))
