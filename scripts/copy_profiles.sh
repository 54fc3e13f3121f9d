# copies the summaries of gpurun_out/<round>/ (written by gpu_profile_round.sh on the GPU box) into profiles/:  bash scripts/copy_profiles.sh r02
R=${1:-r05}; S=gpurun_out/$R; D=profiles
for f in $S/bench_*.json; do [ -s $f ] && cp $f $D/${R}_$(basename $f); done
for wl in cfg2 cfg3 cfg4shard; do for a in i8 fp64; do
  f=$(ls $S/trace_${wl}_$a/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/${R}_bench_${wl}_${a}_kernel_stats.csv
done; done
for t in trace_cfg3_i8 trace_cfg3_fp64 trace_cfg4shard_i8; do [ -s $S/${t}_by_grid.csv ] && cp $S/${t}_by_grid.csv $D/${R}_bench_${t#trace_}_kernel_by_grid.csv; done
cp $S/pmc_summary.json $D/${R}_pmc_summary.json; cp $S/pmc_summary_i8.json $D/${R}_pmc_summary_i8.json
cp $S/mfma_i8_microbench.txt $D/${R}_mfma_i8_microbench.txt; cp $S/mfma_f64_microbench.txt $D/${R}_mfma_f64_microbench.txt
[ -s $S/power_sample.txt ] && cp $S/power_sample.txt $D/${R}_power_sample.txt
for f in ortho_slice_cost.txt partial_download.txt cli_e2e_cfg3.txt solve_profiles.txt k4_bench.txt fp_apply_bench.txt missing_routes.txt; do [ -s $S/$f ] && cp $S/$f $D/${R}_$f; done
ls -la $D | tail -40
