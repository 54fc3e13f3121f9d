import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import flashpca_amd as fp
pre = sys.argv[1]
t0 = time.perf_counter(); N = fp.count_fam_rows(pre + ".fam"); t1 = time.perf_counter()
ctx = fp.Context.from_bed(pre + ".bed", N); t2 = time.perf_counter()
ctx.stats(); t3 = time.perf_counter()
r = ctx.pca(ndim=20, do_loadings=True); t4 = time.perf_counter()
print("fam %.3f s | from_bed (read + H2D) %.3f s = %.2f GB/s | stats %.3f | pca+loadings %.3f (inside %.3f)" % (
    t1 - t0, t2 - t1, os.path.getsize(pre + ".bed") / (t2 - t1) / 1e9, t3 - t2, t4 - t3, r["info"]["seconds_total"]))
