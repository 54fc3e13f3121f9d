# one SQ/GRBM counter pass over a 3-step bench of the default mode at a given width:  bash scripts/gpu_pmc_quick.sh [blockvec] [workload]
B=${1:-0}; WL=${2:-cfg3}; R=gpurun_out/pmcq; rm -rf $R; mkdir -p $R; export TMPDIR=/tmp
PMC_SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $PMC_SQ --output-format csv -d $R/pmc_sq_${WL}_i8 -o pmc -- python bench.py --workload $WL --blockvec $B --steps 3 --warmup 1 --no-cpu-baseline --no-pca --no-alt --no-e2e --traffic none > /dev/null 2>&1
python scripts/summarise_pmc.py $R _i8 | python -c "
import json,sys; d=json.load(sys.stdin)
for wl,v in d.items():
    for k,x in v.items():
        if 'gemm' in k: print(wl,k,{a:round(b,4) for a,b in x.items()})"
find $R -name "*.csv" -size +2M -delete
