#!/bin/bash
# PMC pass over the int8 GEMMs (cfg3) and the i8 peak microbench: clock, MFMA busy, VALU busy, LDS conflicts
mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/run_i8.py <<'PY'
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import flashpca_amd as fp
L = fp.lib()
t = C.c_double()
L.fpca_debug_mfma_peak(1, 20000, 11, C.byref(t)); print("peak random 1w", t.value)
L.fpca_debug_mfma_peak(1, 20000, 10, C.byref(t)); print("peak zero 1w", t.value)
ctx = fp.Context.synthetic(500000, 100000, n_pop=40, accum="i8")
r = ctx.bench_apply(b=32, steps=2, warmup=1)
print(r)
PY
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/i8pmc_$tag -o pmc -- python /tmp/run_i8.py > gpurun_out/i8pmc_$tag.out 2> gpurun_out/i8pmc_$tag.err
done
python - <<'PY' | tee gpurun_out/i8_pmc_summary.txt
import csv,glob,collections
agg=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/i8pmc_*/*counter_collection.csv")):
    per=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "gemm_i8" in n or "mfma_i8_peak" in n:
            key=("K3i" if "<true>" in n else "K2i" if "<false>" in n else "peak")+":"+r["Dispatch_Id"]
            d=per.setdefault(key,{"dur":int(r["End_Timestamp"])-int(r["Start_Timestamp"])})
            d[r["Counter_Name"]]=d.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    for k,v in per.items():
        if v["dur"]<2e6: continue
        name=k.split(":")[0]
        a=agg.setdefault(name,{})
        for c,val in v.items():
            a.setdefault(c,[]).append(val)
for name,a in agg.items():
    print(name, {c: sum(v)/len(v) for c,v in a.items()})
PY
