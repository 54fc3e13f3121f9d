# round 4: is the int8 GEMM's genotype side bound by the texture addresser?  TA / TCP / SQ-VMEM counters of the GEMM kernel,
# headline instance (default) and the 2-tile instance of the cheap passes (--accum i8x4).   bash scripts/gpu_pmc_ta.sh
export TMPDIR=/tmp; R=gpurun_out/pmc_ta; rm -rf $R; mkdir -p $R
rocprofv3 -L 2>/dev/null | grep -oE "\bTCP_[A-Z0-9_a-z]+" | sort -u > $R/tcp_counters.txt
# (few counters per pass, every pass under its own time limit: a counter set the profiler cannot schedule aborts the child and
#  leaves rocprofv3 waiting)
A="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
B="GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_BUSY_avr"
C="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"
D="GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
E="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
F="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL"
for acc in i8 i8x4; do for p in A B C D E F; do
  eval CNT=\$$p
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/${acc}_$p -o pmc -- python bench.py --accum $acc --steps 2 --warmup 1 --no-cpu-baseline --no-pca --no-alt --no-e2e --traffic none > /dev/null 2> $R/${acc}_$p.err
done; done
python - <<'PY'
import csv, glob, collections
for acc in ("i8", "i8x4"):
    for p in "ABCDEF":
        fs = glob.glob("gpurun_out/pmc_ta/%s_%s/**/*counter_collection.csv" % (acc, p), recursive=True)
        if not fs:
            print(acc, p, "no counter file"); continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "k_gemm_i8" in r["Kernel_Name"]]
        agg = collections.defaultdict(list); dur = []
        for r in rows:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        g = sum(agg["GRBM_GUI_ACTIVE"]) / max(1, len(agg["GRBM_GUI_ACTIVE"]))
        print(acc, p, "launches %d  dur %.3f ms  cycles/XCD %.3e" % (len(dur), sum(dur) / max(1, len(dur)) * 1e-6, g / 8))
        for k, v in sorted(agg.items()):
            if k != "GRBM_GUI_ACTIVE": print("     %-40s %.4e   per XCD-cycle %.4f" % (k, sum(v) / len(v), sum(v) / len(v) / (g / 8)))
PY
find $R -name "*.csv" -size +1M -delete
