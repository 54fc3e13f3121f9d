# Regenerates the committed profile evidence of a round:  bash scripts/gpu_profile_round.sh r01
# (run through gpurun; writes under gpurun_out/, copy the summaries into profiles/)
R=${1:-r01}
mkdir -p gpurun_out/$R; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > gpurun_out/$R/bench_cfg2_n1.json 2> gpurun_out/$R/bench_cfg2_n1.err
python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$R/bench_cfg3_n1.json 2>/dev/null
python bench.py --accum fp32 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench_cfg2_n1_fp32.json 2>/dev/null
python bench.py --workload cfg3 --accum fp32 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$R/bench_cfg3_n1_fp32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace_cfg2 -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace_cfg3 -o bench -- python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for wl in cfg2 cfg3; do
  st=3; [ $wl = cfg3 ] && st=2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc_fetch_$wl -o pmc -- python bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmc_write_$wl -o pmc -- python bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/$R/pmc_sq_$wl -o pmc -- python bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2>&1
done
# exact-integer mode (FPCA_ACCUM_I8(8)): bench lines, kernel trace, PMC passes
python bench.py --accum i8 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench_cfg2_n1_i8.json 2>/dev/null
python bench.py --workload cfg3 --accum i8 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$R/bench_cfg3_n1_i8.json 2>/dev/null
python bench.py --workload cfg3 --accum i8x6 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$R/bench_cfg3_n1_i8x6.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace_cfg3_i8 -o bench -- python bench.py --workload cfg3 --accum i8 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for wl in cfg2 cfg3; do
  st=3; [ $wl = cfg3 ] && st=2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc_fetch_${wl}_i8 -o pmc -- python bench.py --workload $wl --accum i8 --steps $st --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmc_write_${wl}_i8 -o pmc -- python bench.py --workload $wl --accum i8 --steps $st --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/$R/pmc_sq_${wl}_i8 -o pmc -- python bench.py --workload $wl --accum i8 --steps $st --warmup 1 --no-cpu-baseline --no-pca > /dev/null 2>&1
done
python scripts/summarise_pmc.py gpurun_out/$R _i8 > gpurun_out/$R/pmc_summary_i8.json
python scripts/mfma_i8_peak.py > gpurun_out/$R/mfma_i8_microbench.txt 2>&1
python scripts/mfma_peak.py > gpurun_out/$R/mfma_f64_microbench.txt 2>&1
python scripts/summarise_pmc.py gpurun_out/$R > gpurun_out/$R/pmc_summary.json
cat gpurun_out/$R/pmc_summary.json; cat gpurun_out/$R/bench_cfg2_n1.json
