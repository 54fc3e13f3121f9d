set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in cfg2 cfg3; do
  steps=3; [ $wl = cfg3 ] && steps=2
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmcA_$wl -o pmc -- python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2> gpurun_out/pmcA_$wl.err
done
python - <<'PY'
import csv,glob,collections
for wl in ("cfg2","cfg3"):
    f=glob.glob("gpurun_out/pmcA_%s/*counter_collection.csv"%wl)
    if not f: print("no file",wl); continue
    agg=collections.defaultdict(list); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:34]
        agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in sorted(set(x[0] for x in agg)):
        if "k_xt_b" in k or "k_x_t" in k:
            g=lambda c: sum(agg[(k,c)])/max(1,len(agg[(k,c)]))
            d=sum(dur[k])/max(1,len(dur[k])) if dur[k] else 0
            print(wl,k,"dur_us %.1f"%(d/1e3),"clkGHz %.3f"%(g("GRBM_GUI_ACTIVE")/8/max(d,1)),"mfma_busy_frac %.3f"%(g("SQ_VALU_MFMA_BUSY_CYCLES")/1024/(g("GRBM_GUI_ACTIVE")/8)),
                  "wave_cyc %.3g wait_inst %.3g wait_any %.3g active %.3g lds_conf %.3g lds_act %.3g"%(g("SQ_WAVE_CYCLES"),g("SQ_WAIT_INST_ANY"),g("SQ_WAIT_ANY"),g("SQ_ACTIVE_INST_ANY"),g("SQ_LDS_BANK_CONFLICT"),g("SQ_LDS_IDX_ACTIVE")))
PY
head -3 gpurun_out/pmcA_cfg2/*counter_collection.csv | cut -c1-600
