set -x
mkdir -p gpurun_out
python - <<'PY'
import ctypes as C, flashpca_amd as fp
L=fp.lib()
for w in (1,2,4):
    t=C.c_double()
    rc=L.fpca_debug_mfma_peak(w, 20000, C.byref(t)); print("mfma f64 peak, %d wave(s)/SIMD: rc=%d %.2f TFLOP/s"%(w,rc,t.value))
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
# PMC passes (separate from tracing): HBM traffic of the two GEMM kernels at cfg2 and cfg3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2> gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2> gpurun_out/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch3 -o pmc -- python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2> gpurun_out/pmc_fetch3.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_mfma -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2> gpurun_out/pmc_mfma.err
ls gpurun_out/pmc_fetch gpurun_out/pmc_mfma; tail -3 gpurun_out/pmc_mfma.err
python - <<'PY'
import csv,glob,collections
for d in ("pmc_fetch","pmc_write","pmc_fetch3","pmc_mfma"):
    for f in glob.glob("gpurun_out/%s/*counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            if "k_xt_b" in k[0] or "k_x_t" in k[0] or "bed_stats" in k[0] or "reduce" in k[0]:
                print(d,k,len(v),sum(v)/len(v))
PY
