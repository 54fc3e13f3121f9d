# Regenerates the committed profile evidence of a round:  bash scripts/gpu_profile_round.sh r02
# (run through gpurun; writes under gpurun_out/<round>/, scripts/copy_profiles.sh copies the summaries into profiles/)
R=${1:-r05}
mkdir -p gpurun_out/$R; export TMPDIR=/tmp; export PYTHONPATH=$PWD
PMC_SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
Q="--no-cpu-baseline --no-pca-hard --no-e2e"
# the default line = the driver's command (500k x 100k headline, exact-integer mode; carries the fp64 kernels' numbers as
# fp64_mode, the CPU baseline and both PCA solves), then the other workloads and the explicit modes
python bench.py > gpurun_out/$R/bench_cfg3_n1.json 2> gpurun_out/$R/bench_cfg3_n1.err
python bench.py --workload cfg2 > gpurun_out/$R/bench_cfg2_n1.json 2>/dev/null
for a in fp64 fp32 i8x4; do
  python bench.py --workload cfg2 --accum $a $Q --no-alt > gpurun_out/$R/bench_cfg2_n1_$a.json 2>/dev/null
  python bench.py --workload cfg3 --accum $a $Q --no-alt > gpurun_out/$R/bench_cfg3_n1_$a.json 2>/dev/null
done
for wl in cfg4shard cfg5shard; do python bench.py --workload $wl $Q > gpurun_out/$R/bench_${wl}_n1.json 2>/dev/null; done
python bench.py --workload cfg4shard --accum i8x4 $Q --no-alt > gpurun_out/$R/bench_cfg4shard_n1_i8x4.json 2>/dev/null   # the cheap passes' kernels on the shard
python bench.py --workload cfg5shard --accum fp32 $Q --no-alt > gpurun_out/$R/bench_cfg5shard_n1_fp32.json 2>/dev/null
for a in i8 fp64; do
  # kernel trace of the driver's command line itself for the default mode (same flags), of --accum fp64 for the other
  if [ $a = i8 ]; then X=""; else X="--accum fp64 $Q --no-alt"; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace_cfg3_$a -o bench -- python bench.py $X > /dev/null 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace_cfg2_$a -o bench -- python bench.py --workload cfg2 --accum $a $Q --no-alt > /dev/null 2>&1
  for wl in cfg2 cfg3; do
    st=3; [ $wl = cfg3 ] && st=2
    for kind in fetch write sq; do
      case $kind in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="--kernel-trace $PMC_SQ";; esac
      if [ $kind = sq ]; then
        timeout 400 rocprofv3 --kernel-trace --pmc $PMC_SQ --output-format csv -d gpurun_out/$R/pmc_${kind}_${wl}_$a -o pmc -- python bench.py --workload $wl --accum $a --steps $st --warmup 1 --no-cpu-baseline --no-pca --no-alt --no-e2e --traffic none > /dev/null 2>&1
      else
        timeout 400 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$R/pmc_${kind}_${wl}_$a -o pmc -- python bench.py --workload $wl --accum $a --steps $st --warmup 1 --no-cpu-baseline --no-pca --no-alt --no-e2e --traffic none > /dev/null 2>&1
      fi
    done
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace_cfg4shard_i8 -o bench -- python bench.py --workload cfg4shard $Q --no-alt > /dev/null 2>&1
python scripts/summarise_pmc.py gpurun_out/$R _fp64 > gpurun_out/$R/pmc_summary.json
python scripts/summarise_pmc.py gpurun_out/$R _i8 > gpurun_out/$R/pmc_summary_i8.json
python scripts/mfma_i8_peak.py > gpurun_out/$R/mfma_i8_microbench.txt 2>&1
python scripts/mfma_peak.py > gpurun_out/$R/mfma_f64_microbench.txt 2>&1
python bench.py --workload cfg5 $Q --no-e2e > gpurun_out/$R/bench_cfg5_n1.json 2>/dev/null
python bench.py --blockvec 32 $Q --no-e2e --no-alt > gpurun_out/$R/bench_cfg3_n1_b32.json 2>/dev/null
python scripts/ortho_slice_cost.py > gpurun_out/$R/ortho_slice_cost.txt 2>&1
python scripts/partial_download_probe.py > gpurun_out/$R/partial_download.txt 2>&1
python scripts/solve_profiles.py 1 2 > gpurun_out/$R/solve_profiles.txt 2>&1
python scripts/k4_bench.py > gpurun_out/$R/k4_bench.txt 2>&1
python scripts/fp_apply_bench.py > gpurun_out/$R/fp_apply_bench.txt 2>&1
python scripts/missing_routes_probe.py > gpurun_out/$R/missing_routes.txt 2>&1
FPCA_TIMING=1 bash scripts/gpu_cli_e2e.sh 500000 100000 3 > gpurun_out/$R/cli_e2e_cfg3.txt 2>&1
bash scripts/power_sample.sh i8 > gpurun_out/$R/power_sample.txt 2>&1
for t in trace_cfg3_i8 trace_cfg3_fp64 trace_cfg4shard_i8; do python scripts/summarise_trace.py gpurun_out/$R/$t > gpurun_out/$R/${t}_by_grid.csv; done
# slim the raw traces before they travel back (the stats CSVs are what profiles/ keeps)
find gpurun_out/$R -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out/$R -name "*counter_collection.csv" -size +8M -delete
cat gpurun_out/$R/pmc_summary_i8.json | head -70; tail -c 1500 gpurun_out/$R/bench_cfg3_n1.json
