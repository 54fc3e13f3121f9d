"""What one rank's share of the result download costs (fpca_pca_opts.partial_rows; input of profiles/r05_scale_model.md): seconds_download
of a k = 20 solve at the heights a rank's row slice has with 1 / 2 / 4 / 8 ranks of a 500,000-sample run (the whole-block download of
round 4 funnelled all 500,000 rows through rank 0)."""
import flashpca_amd as fp

for rows in (500000, 250000, 125000, 62500):
    with fp.Context.synthetic(rows, 4096, n_pop=40, accum="auto") as c:
        c.pca(ndim=20, allow_unconverged=True, max_applies=7)
        best = 1e9
        for _ in range(3):
            r = c.pca(ndim=20, allow_unconverged=True, max_applies=7)
            best = min(best, r["info"]["seconds_download"])
        print("rows %7d: U + Px download %.3f ms (2 x %.1f MB)" % (rows, best * 1e3, rows * 20 * 8 / 1e6), flush=True)
