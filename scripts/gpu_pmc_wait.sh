#!/bin/bash
# where do the int8 GEMM's wave cycles go?  (SQ wait / active breakdown)
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_(WAIT|ACTIVE|INST_LEVEL|INSTS|BUSY|WAVE)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/r02/sq_counters_available.txt
cat gpurun_out/r02/sq_counters_available.txt; echo
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $set GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02/pmc_wait_$tag -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pca --no-alt > /dev/null 2> gpurun_out/r02/pmc_wait_$tag.err
  python - <<PY
import csv,collections,glob
fs=glob.glob('gpurun_out/r02/pmc_wait_$tag/*counter_collection.csv')
if not fs: print('no output for $tag'); print(open('gpurun_out/r02/pmc_wait_$tag.err').read()[-600:])
else:
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'k_gemm_i8' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('%-24s %.4e  (%d launches)'%(k, sum(v)/len(v), len(v)))
PY
done
find gpurun_out/r02 -name "*counter_collection.csv" -size +8M -delete
