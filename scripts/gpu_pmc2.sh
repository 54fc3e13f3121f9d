mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/mfma_peak.py
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmcB -o pmc -- python scripts/mfma_peak.py > /dev/null 2> gpurun_out/pmcB.err
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmcB/*counter_collection.csv")[0]
rows=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_mfma_peak" in r["Kernel_Name"]:
        rows.setdefault(r["Dispatch_Id"],{"dur":int(r["End_Timestamp"])-int(r["Start_Timestamp"]),"grid":r["Grid_Size"]})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in rows.items():
    if v["dur"]>1e6:
        print(k,v["grid"],"dur_ms %.2f"%(v["dur"]/1e6),"clk %.3f GHz"%(v["GRBM_GUI_ACTIVE"]/8/v["dur"]),"busy %.3f"%(v["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(v["GRBM_GUI_ACTIVE"]/8)),"cyc/mfma %.1f"%(v["SQ_VALU_MFMA_BUSY_CYCLES"]/v["SQ_INSTS_MFMA"]))
PY
