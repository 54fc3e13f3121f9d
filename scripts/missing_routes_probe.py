"""Route of the missing-call indicator and block-apply time (b = 16, exact passes) against the per-SNP distribution of the missing
calls at 500,000 x 100,000: uniform rates, log-normal per-SNP rates (sd of ln(rate) 1.0 / 1.5 / 2.0) and the concentrated profile.
usage: python scripts/missing_routes_probe.py"""
import flashpca_amd as fp

N, P, b = 500000, 100000, 16
MODE = {0: "two-matrix", 1: "two-matrix (skip)", 2: "none", 3: "sparse", 4: "hybrid"}
cases = [("uniform 0.1 %", dict(missing_rate=0.001)), ("uniform 0.5 %", dict(missing_rate=0.005)), ("uniform 1 %", dict(missing_rate=0.01)),
         ("uniform 2 %", dict(missing_rate=0.02)), ("concentrated (5 % of the SNPs at 10-30 %)", dict(missing_model=1))]
for mean in (0.005, 0.01, 0.02):
    for sig in (1.0, 1.5, 2.0):
        cases.append(("log-normal mean %.1f %% sigma %.1f" % (100 * mean, sig), dict(missing_rate=mean, missing_model=2, lognormal_sigma=sig)))
for name, kw in cases:
    with fp.Context.synthetic(N, P, n_pop=40, accum="i8", **kw) as c:
        ms, _ = c.stats()
        r = c.bench_apply(b=b, steps=6, warmup=3)
        print("%-46s route %-10s  apply %.2f ms (stages %.2f + %.2f)" % (name, MODE[c.missing_mode(b)], r["ms_xt"] + r["ms_x"], r["ms_xt"], r["ms_x"]), flush=True)
